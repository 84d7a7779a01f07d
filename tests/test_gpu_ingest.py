"""GPU tests of the callers either side of the path (hotstuff_amd/ingest.py) with the real engine."""
import pytest
import torch

import synth
from conftest import oracle_verify_many
from hotstuff_amd import _lib
from hotstuff_amd.crypto import Digest, PublicKey, Signature
from hotstuff_amd.ingest import Timeout, VerifiedQCCache, Vote, VoteIngest
from hotstuff_amd.messages import QC, Committee, InvalidSignature

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def engine():
    assert torch.cuda.is_available()
    _lib.init(0, 8192)
    yield


def _committee_and_votes(n, rnd=3, cache_keys=True):
    h = Digest(bytes([7]) * 32)
    d = Vote(h, rnd, None, None).digest()
    votes = []
    for i in range(n):
        pk, sg = synth.sign(synth.key_seed(5000 + i), d.b)
        votes.append(Vote(h, rnd, PublicKey(pk), Signature.from_bytes(sg)))
    return Committee({v.author: 1 for v in votes}, cache_keys=cache_keys), votes


def test_leader_vote_ingest_one_gpu_call_per_quorum():
    committee, votes = _committee_and_votes(100)                 # quorum 67
    ing = VoteIngest(committee)
    qc = None
    for v in votes:
        r = ing.add_vote(v)
        qc = qc or r
    assert qc is not None and len(qc.votes) == 67 and ing.gpu_calls == 1
    assert qc.verify(committee) is None                          # what every other node will do with it
    for v in votes[:3]:
        assert v.verify(committee) is None                       # the reference's per-vote path agrees


def test_timeout_storm_verifies_the_high_qc_once():
    committee, votes = _committee_and_votes(40, rnd=8)
    high_qc = QC(votes[0].hash, 8, [(v.author, v.signature) for v in votes[:27]])
    cache = VerifiedQCCache()
    timeouts = []
    for i, v in enumerate(votes):
        t = Timeout(high_qc, 9, v.author, None)
        _, sg = synth.sign(synth.key_seed(5000 + i), t.digest().b)
        t.signature = Signature.from_bytes(sg)
        timeouts.append(t)
    for t in timeouts:
        assert t.verify(committee, qc_cache=cache) is None
    assert cache.misses == 1 and cache.hits == len(timeouts) - 1
    forged = QC(high_qc.hash, 8, [high_qc.votes[0], high_qc.votes[1]] + [(a, high_qc.votes[0][1]) for a, _ in high_qc.votes[2:]])
    bad = Timeout(forged, 9, timeouts[0].author, timeouts[0].signature)
    with pytest.raises(InvalidSignature):
        bad.verify(committee, qc_cache=cache)
    with pytest.raises(InvalidSignature):                        # wrong signature on the timeout itself
        Timeout(high_qc, 10, timeouts[0].author, timeouts[0].signature).verify(committee, qc_cache=cache)


def test_ingest_drops_exactly_what_the_oracle_rejects(oracle):
    """SURVEY.md 8f-1 with the real engine: a leader's vote stream with invalid votes injected - flipped
    signature bits, a valid signature under another author's name, forgeries arriving BEFORE the genuine vote of
    the same authority.  The set VoteIngest drops must be the set of votes the oracle's per-signature strict
    verdict rejects (what the reference's Vote::verify would have dropped one by one), and the QC must carry
    at least 2f+1 verified votes of distinct authorities."""
    committee, votes = _committee_and_votes(60, rnd=4, cache_keys=False)        # quorum 41
    stream = []
    for i, v in enumerate(votes):
        if i % 7 == 3:                               # a forgery naming this authority arrives first
            stream.append(Vote(v.hash, v.round, v.author, Signature(bytes([i]) * 32, bytes(32))))
        if i % 11 == 5:                              # this authority's own vote is corrupted in flight
            s = bytearray(v.signature.flatten()); s[40] ^= 4
            stream.append(Vote(v.hash, v.round, v.author, Signature.from_bytes(bytes(s))))
            continue
        if i % 13 == 6:                              # a valid signature of the next authority under this author's name
            stream.append(Vote(v.hash, v.round, v.author, votes[i + 1].signature))
            continue
        stream.append(v)
    d = votes[0].digest().b
    verdict = oracle_verify_many(oracle, d, 0, b"".join(v.author.b for v in stream), b"".join(v.signature.flatten() for v in stream),
                                 len(stream), 1)
    ing = VoteIngest(committee, max_pending=16)
    qc, seen = None, 0
    for v in stream:
        r = ing.add_vote(v)
        seen += 1
        if r is not None:
            qc = r
            break
    assert qc is not None and committee.quorum_threshold() == 41 and 41 <= len(qc.votes) <= 41 + 15     # one flush may verify a few more than 2f+1
    expect_dropped = [stream[i].author for i in range(seen) if not verdict[i]]
    assert sorted(a.b for a in ing.dropped) == sorted(a.b for a in expect_dropped) and expect_dropped
    accepted = {(a.b, s.flatten()) for a, s in qc.votes}
    assert all(verdict[i] for i in range(seen) if (stream[i].author.b, stream[i].signature.flatten()) in accepted)
    assert len({a.b for a, _ in qc.votes}) == len(qc.votes)
    assert qc.verify(committee) is None
