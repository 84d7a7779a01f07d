"""Callers either side of the verification path, mirrored so that
their batching logic can be exercised and measured; protocol-neutral.

* ``VoteIngest`` - the leader's vote path.  The reference verifies every incoming vote singly
  on the Core task (``Core::handle_vote`` consensus/src/core.rs:203-225 -> ``Vote::verify``
  messages.rs:136-146 -> one strict verification) and only then aggregates
  (``QCMaker::append`` aggregator.rs:74-94): ~2f+1 serial verifications per round.  Here the
  structural checks run at arrival, signatures wait, and one GPU call verifies everything
  pending as soon as the pending + verified stake could form a quorum.  Invalid votes are
  dropped individually (the per-signature bitmap says which), exactly the set the reference
  would have dropped one by one; the QC is emitted once, like ``QCMaker``.

* ``VerifiedQCCache`` - ``Timeout::verify`` re-verifies the embedded ``high_qc`` for every
  timeout received (messages.rs:261-263, core.rs:227-238): up to N identical batch
  verifications of 2f+1 signatures per view change.  Remembering the fingerprints of QCs that
  verified makes that O(N) instead of O(N^2).

The verifier is injectable (``verify_fn(digest: Digest, votes: [(PublicKey, Signature)]) ->
[bool]``, strict per-signature verdicts); the default is the GPU engine.
"""
import hashlib
import struct
from collections import OrderedDict

from .crypto import Digest, Signature
from .messages import QC, AuthorityReuse, InvalidSignature, UnknownAuthority


def _gpu_verify(committee):
    def verify(digest, votes):
        return Signature.verify_many([digest] * len(votes), votes, keyset=committee.keyset)
    return verify


class Vote:
    """messages.rs:112-118 (hash, round, author, signature)."""

    def __init__(self, hash, round, author, signature):
        self.hash, self.round, self.author, self.signature = hash, round, author, signature

    def digest(self):
        # messages.rs:149-156: SHA-512(hash || round_le)[..32]
        return Digest(hashlib.sha512(self.hash.b + struct.pack("<Q", self.round)).digest()[:32])

    def verify(self, committee):
        """messages.rs:136-146: stake check, then one strict verification."""
        if committee.stake(self.author) <= 0:
            raise UnknownAuthority(self.author)
        try:
            self.signature.verify(self.digest(), self.author)
        except Exception as e:
            raise InvalidSignature() from e


class VoteIngest:
    """Aggregates votes for one (round, block hash) and verifies them in batches."""

    def __init__(self, committee, verify_fn=None, max_pending=None):
        self.committee = committee
        self.verify_fn = verify_fn or _gpu_verify(committee)
        self.max_pending = max_pending
        self.used = set()
        self.pending = []          # votes whose signature has not been checked yet
        self.valid = []            # (author, signature) with a verified signature
        self.valid_weight = 0
        self.done = False
        self.hash = None
        self.round = None
        self.gpu_calls = 0
        self.dropped = []

    def add_vote(self, vote):
        """Structural checks now (aggregator.rs:79-82, messages.rs:138-142), signature later.
        Returns a QC the first time the verified stake reaches the quorum, else None."""
        if self.hash is None:
            self.hash, self.round = vote.hash, vote.round
        elif (vote.hash, vote.round) != (self.hash, self.round):
            raise ValueError("VoteIngest handles one (hash, round); use one instance per aggregator entry")
        if vote.author in self.used:
            raise AuthorityReuse(vote.author)
        if self.committee.stake(vote.author) <= 0:
            raise UnknownAuthority(vote.author)
        self.used.add(vote.author)
        if self.done:
            return None            # the certificate exists; late votes need no verification
        self.pending.append(vote)
        pend_weight = sum(self.committee.stake(v.author) for v in self.pending)
        if (self.valid_weight + pend_weight >= self.committee.quorum_threshold()
                or (self.max_pending and len(self.pending) >= self.max_pending)):
            return self.flush()
        return None

    def flush(self):
        """Verify everything pending in one call; keep the valid votes."""
        if self.pending:
            votes = [(v.author, v.signature) for v in self.pending]
            oks = self.verify_fn(self.pending[0].digest(), votes)
            self.gpu_calls += 1
            for v, ok in zip(self.pending, oks):
                if ok:
                    self.valid.append((v.author, v.signature))
                    self.valid_weight += self.committee.stake(v.author)
                else:
                    self.dropped.append(v.author)     # the reference warns and drops (core.rs:434-439)
            self.pending = []
        if not self.done and self.valid_weight >= self.committee.quorum_threshold():
            self.done = True                          # QCMaker: "ensures QC is only made once"
            return QC(self.hash, self.round, list(self.valid))
        return None


class VerifiedQCCache:
    """Fingerprints of QCs whose signatures verified, bounded LRU."""

    def __init__(self, capacity=1024):
        self.capacity = capacity
        self._seen = OrderedDict()
        self.hits = 0
        self.misses = 0

    @staticmethod
    def fingerprint(qc):
        h = hashlib.sha512()
        h.update(qc.hash.b)
        h.update(struct.pack("<QQ", qc.round, len(qc.votes)))
        for name, sig in qc.votes:
            h.update(name.b)
            h.update(sig.flatten())
        return h.digest()[:32]

    def verify(self, qc, committee):
        """QC::verify with memoisation: the structural checks and the batch verification run
        once per distinct QC (same hash, round and vote list, byte for byte)."""
        fp = self.fingerprint(qc)
        if fp in self._seen:
            self._seen.move_to_end(fp)
            self.hits += 1
            return
        self.misses += 1
        qc.verify(committee)                 # raises on failure: failures are not cached
        self._seen[fp] = True
        if len(self._seen) > self.capacity:
            self._seen.popitem(last=False)


class Timeout:
    """messages.rs:211-216 (high_qc, round, author, signature)."""

    def __init__(self, high_qc, round, author, signature):
        self.high_qc, self.round, self.author, self.signature = high_qc, round, author, signature

    def digest(self):
        # messages.rs:268-275: SHA-512(round_le || high_qc.round_le)[..32]
        return Digest(hashlib.sha512(struct.pack("<QQ", self.round, self.high_qc.round)).digest()[:32])

    def verify(self, committee, qc_cache=None):
        """messages.rs:250-265; with `qc_cache` the embedded high_qc is verified once per
        distinct QC instead of once per timeout."""
        if committee.stake(self.author) <= 0:
            raise UnknownAuthority(self.author)
        try:
            self.signature.verify(self.digest(), self.author)
        except Exception as e:
            raise InvalidSignature() from e
        if self.high_qc.votes or self.high_qc.round != 0 or self.high_qc.hash != Digest():   # != QC::genesis()
            if qc_cache is not None:
                qc_cache.verify(self.high_qc, committee)
            else:
                self.high_qc.verify(committee)
