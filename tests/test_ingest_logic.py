"""CPU-only logic tests of hotstuff_amd/ingest.py (the callers either side of the path,
SURVEY.md 8f rows 1-2) with the ORACLE injected as the verifier."""
import ctypes

import pytest

import synth
from conftest import S
from hotstuff_amd.crypto import Digest, PublicKey, Signature
from hotstuff_amd.ingest import VerifiedQCCache, Vote, VoteIngest
from hotstuff_amd.messages import AuthorityReuse, Committee, UnknownAuthority


def oracle_fn(orc):
    def verify(digest, votes):
        n = len(votes)
        pk = b"".join(k.b for k, _ in votes)
        sg = b"".join(s.flatten() for _, s in votes)
        out = ctypes.create_string_buffer(n)
        orc.hs_oracle_verify_many(digest.b, S(32), S(0), pk, sg, S(n), 1, out, 1)
        return [bool(b) for b in out.raw]
    return verify


def make_votes(n, hash_=Digest(bytes(range(32))), rnd=5):
    proto = Vote(hash_, rnd, None, None)
    d = proto.digest()
    votes = []
    for i in range(n):
        pk, sg = synth.sign(synth.key_seed(1000 + i), d.b)
        votes.append(Vote(hash_, rnd, PublicKey(pk), Signature.from_bytes(sg)))
    return votes


def test_vote_ingest_batches_until_quorum_is_possible(oracle):
    votes = make_votes(7)
    committee = Committee({v.author: 1 for v in votes})          # N = 7 -> quorum 5
    ing = VoteIngest(committee, verify_fn=oracle_fn(oracle))
    for v in votes[:4]:
        assert ing.add_vote(v) is None
    assert ing.gpu_calls == 0                                     # nothing verified yet: no quorum possible
    qc = ing.add_vote(votes[4])
    assert qc is not None and ing.gpu_calls == 1 and len(qc.votes) == 5
    assert (qc.hash, qc.round) == (votes[0].hash, votes[0].round)
    assert ing.add_vote(votes[5]) is None                         # QC is made once (aggregator.rs:87)
    with pytest.raises(AuthorityReuse):
        ing.add_vote(votes[0])
    with pytest.raises(UnknownAuthority):
        ing.add_vote(Vote(votes[0].hash, 5, PublicKey(bytes(32)), votes[0].signature))


def test_vote_ingest_drops_exactly_the_invalid_votes(oracle):
    votes = make_votes(7)
    bad = votes[2]
    bad.signature = Signature(bad.signature.part1, bytes([bad.signature.part2[0] ^ 1]) + bad.signature.part2[1:])
    committee = Committee({v.author: 1 for v in votes})
    ing = VoteIngest(committee, verify_fn=oracle_fn(oracle))
    out = [ing.add_vote(v) for v in votes[:5]]
    assert out == [None] * 5 and ing.gpu_calls == 1               # flushed at 5, but only 4 were valid
    assert ing.dropped == [bad.author] and ing.valid_weight == 4
    qc = ing.add_vote(votes[5])
    assert qc is not None and ing.gpu_calls == 2
    assert bad.author not in [a for a, _ in qc.votes] and len(qc.votes) == 5


class _StubQC:
    def __init__(self, tag, ok=True):
        self.hash, self.round, self.votes, self.ok, self.calls = Digest(bytes([tag]) * 32), tag, [], ok, 0

    def verify(self, committee):
        self.calls += 1
        if not self.ok:
            raise ValueError("bad qc")


def test_verified_qc_cache_collapses_a_timeout_storm():
    cache = VerifiedQCCache(capacity=2)
    qc = _StubQC(1)
    for _ in range(100):                    # N timeouts carrying the same high_qc (core.rs:227-238)
        cache.verify(qc, None)
    assert qc.calls == 1 and cache.hits == 99 and cache.misses == 1
    bad = _StubQC(2, ok=False)
    for _ in range(3):
        with pytest.raises(ValueError):
            cache.verify(bad, None)
    assert bad.calls == 3                   # failures are never cached
    cache.verify(_StubQC(3), None)
    cache.verify(_StubQC(4), None)          # evicts the oldest fingerprint
    cache.verify(qc, None)
    assert qc.calls == 2


def test_forged_votes_cannot_lock_out_an_authority(oracle):
    """anyone can send a garbage-signature vote naming authority X.  The reference
    verifies first and only then marks X used (core.rs:210, aggregator.rs:79-82), so the forgery is
    dropped and X's genuine vote still counts - while the forgery is pending AND after it was dropped."""
    votes = make_votes(4)
    committee = Committee({v.author: 1 for v in votes})          # N = 4 -> quorum 3
    ing = VoteIngest(committee, verify_fn=oracle_fn(oracle))
    forged = [Vote(v.hash, v.round, v.author, Signature(bytes([7]) * 32, bytes(32))) for v in votes]
    # forgeries for three authorities arrive first: together they "could" form a quorum -> one flush drops all three
    out = [ing.add_vote(f) for f in forged[:3]]
    assert out == [None] * 3 and ing.gpu_calls == 1 and len(ing.dropped) == 3 and not ing.used
    # the genuine votes of the same authorities are still welcome (no AuthorityReuse) ...
    assert ing.add_vote(votes[0]) is None
    # ... also while another forgery for that authority is waiting
    assert ing.add_vote(Vote(votes[0].hash, votes[0].round, votes[1].author, Signature(bytes([9]) * 32, bytes(32)))) is None
    assert ing.add_vote(votes[1]) is None
    assert ing.add_vote(votes[1]) is None                         # byte-identical duplicate of a waiting candidate: ignored
    qc = ing.add_vote(votes[2])
    assert qc is not None and sorted(a.b for a, _ in qc.votes) == sorted(v.author.b for v in votes[:3])
    assert len(qc.votes) == 3 and len({a.b for a, _ in qc.votes}) == 3
    # only now - after a signature from it verified - is the author "used"
    with pytest.raises(AuthorityReuse):
        ing.add_vote(forged[0])


def test_timeout_genesis_check_compares_hash_and_round_only():
    """`high_qc != QC::genesis()` uses QC's PartialEq = (hash, round) (messages.rs:217-221): a timeout whose
    high_qc has hash 0 / round 0 skips the QC verification even when it carries votes."""
    from hotstuff_amd.ingest import Timeout
    from hotstuff_amd.messages import QC

    class _Sig:
        def verify(self, d, a):
            return None
    k = PublicKey(bytes([1]) * 32)
    committee = Committee({k: 1})
    qc = QC(Digest(), 0, [(k, Signature())])          # would fail verification if it were looked at
    Timeout(qc, 3, k, _Sig()).verify(committee)        # must not raise
    qc2 = QC(Digest(bytes([1]) * 32), 0, [])
    with pytest.raises(Exception):
        Timeout(qc2, 3, k, _Sig()).verify(committee)   # not genesis: verified (and rejected: no quorum)


def _garbage(v, tag):
    return Vote(v.hash, v.round, v.author, Signature(bytes([tag % 251 + 1]) * 32, bytes([tag // 251]) + bytes(31)))


def test_forgery_stream_cannot_force_one_call_per_vote(oracle):
    """once verified + pending stake reaches the quorum, every forged vote for a fresh authority
    used to trigger its own verification call.  Now the second miss in a row starts coalescing: forgeries are verified in
    groups of 2, 4, 8 ... and the honest last vote completes the certificate with the group it arrives in, or at the
    deadline (tick)."""
    votes = make_votes(40)
    committee = Committee({v.author: 1 for v in votes})          # N = 40 -> quorum 27
    now = [0.0]
    ing = VoteIngest(committee, verify_fn=oracle_fn(oracle), clock=lambda: now[0], max_wait=0.002)
    for v in votes[:26]:
        assert ing.add_vote(v) is None
    assert ing.gpu_calls == 0
    # 12 forgeries for fresh authorities, one by one, at shortfall 1: each "could" complete the quorum
    calls_before = ing.gpu_calls
    for t, v in enumerate(votes[26:38]):
        assert ing.add_vote(_garbage(v, t)) is None
    forged_calls = ing.gpu_calls - calls_before
    assert forged_calls <= 5, forged_calls                        # 1 + 1 + groups of 2, 4, (4 waiting): not 12
    assert ing.valid_weight == 26
    # the honest last vote: it joins the group that is waiting; the deadline verifies it if nobody else arrives
    qc = ing.add_vote(votes[38])
    if qc is None:
        assert ing.tick() is None                                 # not yet due
        now[0] += 0.003
        qc = ing.tick()
    assert qc is not None and len(qc.votes) == 27 and ing.done
    assert sorted(a.b for a in ing.dropped) == sorted(v.author.b for v in votes[26:38])[:len(ing.dropped)]


def test_same_author_flood_is_bounded_in_memory_and_calls(oracle):
    """unlimited garbage votes naming ONE authority piled up unverified (distinct (author, signature)
    keys, no stake added, nothing flushed).  Now at most `per_author_cap` wait per author: the queue stays bounded and the
    flood costs one call per per_author_cap votes - and the authority's genuine vote still counts afterwards."""
    votes = make_votes(7)
    committee = Committee({v.author: 1 for v in votes})          # quorum 5
    ing = VoteIngest(committee, verify_fn=oracle_fn(oracle), per_author_cap=2)
    peak = 0
    for t in range(100):
        assert ing.add_vote(_garbage(votes[0], t)) is None
        peak = max(peak, len(ing.pending))
    assert peak <= 2 and ing.gpu_calls <= 50 and not ing.used
    assert len(ing.pending) <= 2 and len(ing._pending_keys) == len(ing.pending)
    for v in votes[:4]:
        assert ing.add_vote(v) is None
    qc = ing.add_vote(votes[4])
    assert qc is not None and len(qc.votes) == 5 and votes[0].author in [a for a, _ in qc.votes]


def test_default_max_pending_is_finite(oracle):
    votes = make_votes(4)
    committee = Committee({v.author: 1 for v in votes})
    ing = VoteIngest(committee, verify_fn=oracle_fn(oracle))
    assert ing.max_pending == committee.size() + 64


def test_cap_flush_that_completes_the_quorum_hands_the_certificate_over(oracle):
    """while coalescing, the per-author cap's flush can complete the certificate - for the very
    author whose extra vote triggered it.  add_vote then raised AuthorityReuse (the author had just become `used`) and the
    QC was lost: done = True, every later add_vote / tick returned None, the round stalled until its timeout."""
    votes = make_votes(7)
    committee = Committee({v.author: 1 for v in votes})          # quorum 5
    now = [0.0]
    ing = VoteIngest(committee, verify_fn=oracle_fn(oracle), per_author_cap=2, clock=lambda: now[0], max_wait=1.0)
    for v in votes[:4]:
        assert ing.add_vote(v) is None
    t = iter(range(1000))
    assert ing.add_vote(_garbage(votes[4], next(t))) is None      # could complete, does not: miss 1
    assert ing.add_vote(_garbage(votes[5], next(t))) is None      # miss 2: groups of 2 from here
    assert ing.add_vote(_garbage(votes[4], next(t))) is None
    assert ing.add_vote(_garbage(votes[5], next(t))) is None      # group of 2 verified: miss 3, groups of 4
    assert ing._misses == 3 and ing.valid_weight == 4 and not ing.pending
    assert ing.add_vote(_garbage(votes[6], next(t))) is None
    assert ing.add_vote(votes[6]) is None                         # the genuine vote waits with the forgery (group of 4)
    assert len(ing.pending) == 2 and not ing.done
    qc = ing.add_vote(_garbage(votes[6], next(t)))                # third vote naming author 6: the cap's flush completes the QC
    assert qc is not None and len(qc.votes) == 5 and ing.done
    assert votes[6].author in [a for a, _ in qc.votes]
    with pytest.raises(AuthorityReuse):
        ing.add_vote(votes[6])                                    # and only now is the author "reused"
