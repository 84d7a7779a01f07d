"""Call-site mirror of consensus/src/messages.rs for the verification path.

Only what surrounds the signature checks is restated, so that parity tests read like
the reference's own (consensus/src/tests/messages_tests.rs):

* ``Committee.stake`` / ``quorum_threshold``   consensus/src/config.rs:63-72
* ``QC.digest`` / ``QC.verify``                messages.rs:201-208, 180-198
* ``TC.verify``                                messages.rs:290-315

The structural checks (duplicate authority, unknown authority, quorum) run on the host
exactly in the reference's order and raise before any signature is looked at; the
signature checks go to the GPU engine: ``QC.verify`` -> one shared-digest BATCH call,
``TC.verify`` -> one per-vote-digest STRICT call replacing the reference's serial loop
of ``Signature::verify`` (messages.rs:307-313).
"""
import hashlib
import struct

from .crypto import CryptoError, Digest, Signature


class ConsensusError(Exception):
    """consensus::error::ConsensusError (the variants this path can raise)."""


class AuthorityReuse(ConsensusError):
    pass


class UnknownAuthority(ConsensusError):
    pass


class QCRequiresQuorum(ConsensusError):
    pass


class TCRequiresQuorum(ConsensusError):
    pass


class InvalidSignature(ConsensusError):
    """ConsensusError::InvalidSignature(#[from] CryptoError) consensus/src/error.rs:38-39."""


class Committee:
    """consensus::config::Committee reduced to what verify needs: name -> stake."""

    def __init__(self, authorities, epoch=0, cache_keys=False):
        self.authorities = dict(authorities)   # PublicKey -> stake
        self.epoch = epoch
        self._keyset = None
        self._cache_keys = cache_keys

    @property
    def keyset(self):
        """The committee key cache (hotstuff_amd._lib.KeySet), built on first use when the
        committee was created with cache_keys=True: the committee is static for the epoch, so
        the per-key tables are built once and every QC / TC verification reuses them."""
        if self._cache_keys and self._keyset is None:
            from . import _lib
            self._keyset = _lib.KeySet([k.b for k in self.authorities])
        return self._keyset

    def size(self):
        return len(self.authorities)

    def stake(self, name):
        return self.authorities.get(name, 0)

    def quorum_threshold(self):
        # config.rs:67-72: 2 * total / 3 + 1
        return 2 * sum(self.authorities.values()) // 3 + 1


def _check_quorum(names, committee, err):
    weight = 0
    used = set()
    for name in names:
        if name in used:
            raise AuthorityReuse(name)
        voting_rights = committee.stake(name)
        if voting_rights <= 0:
            raise UnknownAuthority(name)
        used.add(name)
        weight += voting_rights
    if weight < committee.quorum_threshold():
        raise err()


class QC:
    """messages.rs:164-169: hash, round, votes: Vec<(PublicKey, Signature)>."""

    def __init__(self, hash=None, round=0, votes=None):
        self.hash = hash if hash is not None else Digest()
        self.round = round
        self.votes = list(votes or [])

    @classmethod
    def genesis(cls):
        return cls()

    def digest(self):
        # messages.rs:201-208: SHA-512(hash || round_le_u64)[..32]
        h = hashlib.sha512()
        h.update(self.hash.b)
        h.update(struct.pack("<Q", self.round))
        return Digest(h.digest()[:32])

    def verify(self, committee):
        _check_quorum((name for name, _ in self.votes), committee, QCRequiresQuorum)
        try:
            Signature.verify_batch(self.digest(), self.votes, keyset=committee.keyset)
        except CryptoError as e:
            raise InvalidSignature() from e


class TC:
    """messages.rs:283-287: round, votes: Vec<(PublicKey, Signature, Round)>."""

    def __init__(self, round=0, votes=None):
        self.round = round
        self.votes = list(votes or [])

    def vote_digest(self, high_qc_round):
        # messages.rs:307-311: SHA-512(round_le || high_qc_round_le)[..32]
        h = hashlib.sha512()
        h.update(struct.pack("<Q", self.round))
        h.update(struct.pack("<Q", high_qc_round))
        return Digest(h.digest()[:32])

    def verify(self, committee):
        """messages.rs:290-315.  The reference hashes every vote's digest on the host and calls
        Signature::verify n times; here (round, high_qc_rounds) go to the engine, which builds
        the digests on the device and verifies all votes in one call (strict semantics)."""
        from . import _lib
        _check_quorum((name for name, _, _ in self.votes), committee, TCRequiresQuorum)
        if not self.votes:
            return
        ks = committee.keyset
        if ks is not None and not all(a.b in ks.index for a, _, _ in self.votes):
            ks = None
        try:
            ok, _ = _lib.verify_tc_packed(self.round, [r for _, _, r in self.votes],
                                          b"".join(a.b for a, _, _ in self.votes),
                                          b"".join(s.flatten() for _, s, _ in self.votes), keyset=ks)
        except _lib.EngineError as e:
            # an engine fault is not a verdict: the reference's own loop (per-vote digest on the host, strict verification) through
            # the embedder's CPU verifier when there is one (crypto.cpu_verify); InvalidSignature without one, as before
            from . import crypto
            try:
                ok = all(crypto._after_fault(e, b"".join(self.vote_digest(r).b for _, _, r in self.votes), 32,
                                             b"".join(a.b for a, _, _ in self.votes), b"".join(s.flatten() for _, s, _ in self.votes),
                                             len(self.votes), _lib.STRICT))
            except CryptoError as ce:
                raise InvalidSignature() from ce
        if not ok:
            raise InvalidSignature()

    def high_qc_rounds(self):
        return [r for _, _, r in self.votes]
