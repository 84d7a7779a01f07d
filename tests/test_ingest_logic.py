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
