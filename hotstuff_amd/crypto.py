"""Mirror of the reference's crypto types for the verification path.

Same names, argument meaning and error behaviour as /root/reference/crypto/src/lib.rs:
``Digest`` (:21-22), ``PublicKey`` (:65-66), ``Signature {part1, part2}`` (:178-182),
``Signature::flatten`` (:193-198), ``Signature::verify`` (:200-204) and
``Signature::verify_batch`` (:206-219).  Verification goes through the C-ABI
(include/hs_ed25519.h) to the HIP kernels; ``Result<(), CryptoError>`` becomes
"return None or raise CryptoError".  Signing (``Signature::new``, lib.rs:185-191) and
``SignatureService`` stay on the reference's CPU path and are out of scope here.
"""
import base64

from . import _lib


class CryptoError(Exception):
    """crypto::CryptoError (= ed25519::Error, lib.rs:18): opaque, carries no reason."""


# The embedder's CPU verifier: callable(digests, digest_stride, pks, sigs, n, mode) -> list of n bools (mode _lib.STRICT or
# _lib.BATCH, packed bytes as the C ABI takes them) - in the Rust shim the reference's own dalek bodies (INTEGRATION.md
# section 3).  An ENGINE FAULT (_lib.EngineError: no device, a HIP error, a library shut down) is not a verdict: with a CPU
# verifier set every entry point below answers there - never Ok on error, and never Err for a valid certificate because the
# device failed (a TC refused during a view change costs liveness, consensus/src/core.rs:434-439); without one the fault is
# CryptoError, as before.  engine_faults counts the calls answered on the CPU for that reason.
cpu_verify = None
engine_faults = 0


def _after_fault(e, digests, digest_stride, pks, sigs, n, mode):
    global engine_faults
    if cpu_verify is None:
        raise CryptoError(str(e)) from e
    engine_faults += 1
    return list(cpu_verify(bytes(digests), digest_stride, bytes(pks), bytes(sigs), n, _lib.BATCH if mode == _lib.BATCH_RLC else mode))


def _fixed(b, n, what):
    b = bytes(b)
    if len(b) != n:
        raise ValueError("%s must be %d bytes" % (what, n))
    return b


class Digest:
    """crypto::Digest: 32 bytes (lib.rs:21-22)."""
    __slots__ = ("b",)

    def __init__(self, b=bytes(32)):
        self.b = _fixed(b, 32, "Digest")

    def to_vec(self):
        return self.b

    def __eq__(self, o):
        return isinstance(o, Digest) and self.b == o.b

    def __hash__(self):
        return hash(self.b)

    def __repr__(self):
        return base64.b64encode(self.b).decode()


class PublicKey:
    """crypto::PublicKey: 32-byte compressed Edwards point (lib.rs:65-66)."""
    __slots__ = ("b",)

    def __init__(self, b=bytes(32)):
        self.b = _fixed(b, 32, "PublicKey")

    def encode_base64(self):
        return base64.b64encode(self.b).decode()

    @classmethod
    def decode_base64(cls, s):
        return cls(base64.b64decode(s))

    def __eq__(self, o):
        return isinstance(o, PublicKey) and self.b == o.b

    def __hash__(self):
        return hash(self.b)

    def __repr__(self):
        return self.encode_base64()


class Signature:
    """crypto::Signature {part1: R, part2: s} (lib.rs:178-182)."""
    __slots__ = ("part1", "part2")

    def __init__(self, part1=bytes(32), part2=bytes(32)):
        self.part1 = _fixed(part1, 32, "Signature.part1")
        self.part2 = _fixed(part2, 32, "Signature.part2")

    @classmethod
    def from_bytes(cls, b):
        b = _fixed(b, 64, "Signature")
        return cls(b[:32], b[32:])

    @classmethod
    def default(cls):
        """Signature::default(): 64 zero bytes (used by crypto_tests.rs:111)."""
        return cls()

    def flatten(self):
        return self.part1 + self.part2

    def verify(self, digest, public_key):
        """lib.rs:200-204: strict single verification.  Raises CryptoError on reject
        or on any engine failure (never reports success on error)."""
        try:
            ok, _ = _lib.verify_packed(digest.b, public_key.b, self.flatten(), 1, _lib.STRICT, shared=True)
        except _lib.EngineError as e:
            ok = all(_after_fault(e, digest.b, 0, public_key.b, self.flatten(), 1, _lib.STRICT))
        if not ok:
            raise CryptoError()

    @staticmethod
    def verify_batch(digest, votes, keyset=None):
        """lib.rs:206-219: one shared digest, iterable of (PublicKey, Signature);
        empty -> Ok.  All-or-nothing.  With `keyset` (the committee key cache) and every
        key registered in it, the cached kernels are used; the verdict is the same."""
        pks = bytearray()
        sigs = bytearray()
        keys = []
        n = 0
        for key, sig in votes:
            pks += key.b
            keys.append(key.b)
            sigs += sig.part1
            sigs += sig.part2
            n += 1
        if n == 0:
            return
        try:
            if keyset is not None and all(k in keyset.index for k in keys):
                packed, _ = keyset.indices(keys)
                ok, _ = keyset.verify_packed(digest.b, packed, bytes(sigs), n, _lib.BATCH_RLC, shared=True)
            else:
                # BATCH_RLC = dalek's verify_batch as computed (one random linear combination, MSM on the GPU) from
                # hs_ed25519_rlc_min_batch() signatures up, the per-signature kernels below it and as its localiser
                ok, _ = _lib.verify_packed(digest.b, bytes(pks), bytes(sigs), n, _lib.BATCH_RLC, shared=True)
        except _lib.EngineError as e:
            ok = all(_after_fault(e, digest.b, 0, pks, sigs, n, _lib.BATCH))
        if not ok:
            raise CryptoError()

    @staticmethod
    def verify_many(digests, votes, keyset=None):
        """One digest per (PublicKey, Signature) pair, strict semantics - the batched
        form of the loop at consensus/src/messages.rs:307-313.  Returns the per-pair
        accept list (the reference stops at the first failure; "all accepted" is the
        same predicate)."""
        votes = list(votes)
        n = len(votes)
        if n == 0:
            return []
        pks = b"".join(k.b for k, _ in votes)
        sigs = b"".join(s.flatten() for _, s in votes)
        dg = b"".join(d.b for d in digests)
        try:
            if keyset is not None and all(k.b in keyset.index for k, _ in votes):
                packed, _ = keyset.indices([k.b for k, _ in votes])
                _, bm = keyset.verify_packed(dg, packed, sigs, n, _lib.STRICT, shared=False, want_bitmap=True)
            else:
                _, bm = _lib.verify_packed(dg, pks, sigs, n, _lib.STRICT, shared=False, want_bitmap=True)
        except _lib.EngineError as e:
            return [bool(v) for v in _after_fault(e, dg, 32, pks, sigs, n, _lib.STRICT)]
        return [bool(bm[i >> 3] >> (i & 7) & 1) for i in range(n)]
