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
    """Aggregates votes for one (round, block hash) and verifies them in batches.

    Authority bookkeeping follows the reference's ORDER: there a vote is verified first
    (core.rs:210 Vote::verify) and only a vote that verified marks its author used
    (aggregator.rs:79-82 QCMaker::append).  So here an author is `used` only once a signature
    from it has verified.  Unverified candidates wait keyed by (author, signature): several may
    wait for the same author (anyone can send a garbage-signature vote naming authority X; it
    must neither block X's genuine vote while pending nor after it has been dropped), an exact
    duplicate of a waiting candidate is ignored, and at most one candidate per author is
    accepted.  AuthorityReuse is raised exactly where the reference raises it: for a vote
    whose author already has an accepted vote.

    What a hostile peer can cost (the reference verifies each vote on arrival and holds no queue, so it has neither
    problem):

    * MEMORY: at most `per_author_cap` candidates wait per author; the candidate that would exceed it triggers a
      verification of everything pending first (which settles that author's candidates) - so the queue never exceeds
      per_author_cap x committee size, `max_pending` (default: committee size + 64) flushes before that, and a flood
      naming ONE authority costs one call per `per_author_cap` votes.  The stake that could still verify is kept
      incrementally (`_pend_weight`), not recomputed per vote.
    * CALLS: once verified + pending stake reaches the quorum a verification could complete the certificate, so it
      runs at once - the honest case, and the case of one or two bad votes in the round.  But a peer streaming
      forgeries for distinct authorities would then buy one GPU call (0.1 - 0.4 ms, against 0.04 ms on a CPU core for
      the single verification the reference does) per vote.  From the second verification in a row that COULD have
      completed the quorum and did not, the next one waits until 2, 4, 8 ... `max_coalesce` votes are pending
      - or until the oldest has waited `max_wait` seconds (`tick`, driven by the caller's event loop: the
      reference's Core has a timer anyway, core.rs:422-433) - so forgeries are verified in growing groups and an
      honest last vote is delayed by `max_wait` at most, and only while the round is under that attack.
    """

    def __init__(self, committee, verify_fn=None, max_pending=None, per_author_cap=2, max_coalesce=64, max_wait=0.002,
                 clock=None):
        import time
        self.committee = committee
        self.verify_fn = verify_fn or _gpu_verify(committee)
        self.max_pending = max_pending if max_pending else committee.size() + 64
        self.per_author_cap = max(1, per_author_cap)
        self.max_coalesce = max(1, max_coalesce)
        self.max_wait = max_wait
        self.clock = clock or time.monotonic
        self._quorum = committee.quorum_threshold()   # once: the committee sums its stakes per call (config.rs:67-72), and asking per
                                                      # vote (aggregator.rs:85 does) is 66 M steps for the 6,667 votes of N = 10,000
        self.used = set()          # authors with a VERIFIED vote (aggregator.rs `used`)
        self.pending = []          # votes whose signature has not been checked yet
        self._pending_keys = set() # (author bytes, signature bytes) of the waiting candidates
        self._pend_count = {}      # author -> candidates waiting
        self._pend_weight = 0      # stake that COULD verify: one candidate per author at most
        self._oldest = None        # arrival time of the oldest waiting candidate
        self._misses = 0           # verifications in a row that could have completed the quorum and did not
        self.valid = []            # (author, signature) with a verified signature
        self.valid_weight = 0
        self.done = False
        self.hash = None
        self.round = None
        self.gpu_calls = 0
        self.dropped = []

    def _could_complete(self):
        return self.valid_weight + self._pend_weight >= self._quorum

    def _group_size(self):
        """votes that must be waiting before the next quorum-completing attempt (1 until two attempts in a row failed)"""
        return 1 if self._misses < 2 else min(self.max_coalesce, 1 << (self._misses - 1))

    def add_vote(self, vote):
        """Structural checks now (messages.rs:138-142, aggregator.rs:79-82), signature later.
        Returns a QC the first time the verified stake reaches the quorum, else None."""
        if self.hash is None:
            self.hash, self.round = vote.hash, vote.round
        elif (vote.hash, vote.round) != (self.hash, self.round):
            raise ValueError("VoteIngest handles one (hash, round); use one instance per aggregator entry")
        if self.committee.stake(vote.author) <= 0:
            raise UnknownAuthority(vote.author)
        if vote.author in self.used:
            raise AuthorityReuse(vote.author)      # a verified vote of this author is already in (aggregator.rs:79-82)
        if self.done:
            return None            # the certificate exists; late votes need no verification
        key = (vote.author.b, vote.signature.flatten())
        if key in self._pending_keys:
            return None            # byte-identical to a candidate already waiting
        qc = None
        if self._pend_count.get(vote.author, 0) >= self.per_author_cap:
            qc = self.flush()      # settle this author's waiting candidates before queueing another
            if qc is not None or self.done:
                return qc          # the flush completed the certificate: it MUST reach the caller (the certificate is
                                   # made once; raising AuthorityReuse here instead lost it - ADVICE r3)
            if vote.author in self.used:
                raise AuthorityReuse(vote.author)   # one of them verified: this vote is late, as in the reference
        self._pending_keys.add(key)
        self.pending.append(vote)
        c = self._pend_count.get(vote.author, 0)
        self._pend_count[vote.author] = c + 1
        if c == 0:
            self._pend_weight += self.committee.stake(vote.author)
        if self._oldest is None:
            self._oldest = self.clock()
        if len(self.pending) >= self.max_pending:
            return self.flush()
        if self._could_complete() and (len(self.pending) >= self._group_size()
                                       or self.clock() - self._oldest >= self.max_wait):
            return self.flush()
        return None

    def tick(self):
        """Called by the owner's event loop: verifies what is waiting once the oldest candidate has waited `max_wait`
        (only matters while add_vote is coalescing, i.e. under a forgery stream).  Returns a QC or None."""
        if self.pending and not self.done and self._could_complete() and self.clock() - self._oldest >= self.max_wait:
            return self.flush()
        return None

    def flush(self):
        """Verify everything pending in one call; keep the valid votes (first valid candidate per
        author), drop the rest individually - the set the reference drops one by one."""
        could = self._could_complete()
        if self.pending:
            votes = [(v.author, v.signature) for v in self.pending]
            oks = self.verify_fn(self.pending[0].digest(), votes)
            self.gpu_calls += 1
            for v, ok in zip(self.pending, oks):
                if ok and v.author not in self.used:
                    self.used.add(v.author)
                    self.valid.append((v.author, v.signature))
                    self.valid_weight += self.committee.stake(v.author)
                elif not ok:
                    self.dropped.append(v.author)     # the reference warns and drops (core.rs:434-439)
            self.pending = []
            self._pending_keys = set()
            self._pend_count = {}
            self._pend_weight = 0
            self._oldest = None
        if not self.done and self.valid_weight >= self._quorum:
            self.done = True                          # QCMaker: "ensures QC is only made once"
            return QC(self.hash, self.round, list(self.valid))
        if could and not self.done:
            self._misses += 1                         # this attempt could have completed the certificate and did not
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
        # `high_qc != QC::genesis()`: QC's PartialEq compares hash and round only (messages.rs:217-221)
        if self.high_qc.round != 0 or self.high_qc.hash != Digest():
            if qc_cache is not None:
                qc_cache.verify(self.high_qc, committee)
            else:
                self.high_qc.verify(committee)
