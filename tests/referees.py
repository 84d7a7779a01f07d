"""Two Ed25519 verifiers this project did not write, as referees for the oracle's edge-class labels:

* libsodium (crypto_sign_verify_detached, 1.0.18 in this image): rejects s >= L, small-order R and A, non-canonical A,
  then compares the ENCODING of [s]B - [k]A with R's bytes - cofactorless.  It coincides with dalek's `verify_strict` on
  every input one can construct (a non-canonical R or A of large order has an unknown discrete log, so no equation holds
  under it anyway).
* OpenSSL (EVP_DigestVerify): s < L, lenient point decoding, then the same byte comparison and NO small-order checks - the
  per-signature cofactorless law (`verify_batch`'s summand, HS_ED25519_BATCH) wherever R's bytes are the canonical encoding
  of the point they decode to; a non-canonical R (y >= p, or x = 0 with the sign bit: "-0") can never pass a byte
  comparison, while dalek compares POINTS.

Test infrastructure: skip-if-absent, never imported by the product."""
import ctypes
import ctypes.util
import glob
import os

P = 2 ** 255 - 19
_sodium = None
_crypto = None


def libsodium():
    global _sodium
    if _sodium is None:
        cands = [os.environ.get("HS_LIBSODIUM"), ctypes.util.find_library("sodium")]
        cands += sorted(glob.glob("/opt/conda/lib/libsodium.so*")) + sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libsodium.so*"))
        for c in cands:
            if not c:
                continue
            try:
                lib = ctypes.CDLL(c)
                if lib.sodium_init() < 0:
                    continue
                lib.crypto_sign_verify_detached.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_ulonglong, ctypes.c_char_p]
                lib.crypto_sign_verify_detached.restype = ctypes.c_int
                lib.sodium_version_string.restype = ctypes.c_char_p
                _sodium = lib
                break
            except OSError:
                continue
        if _sodium is None:
            _sodium = False
    return _sodium or None


def sodium_verify(msg, pk, sig):
    return libsodium().crypto_sign_verify_detached(sig, msg, len(msg), pk) == 0


def libcrypto():
    global _crypto
    if _crypto is None:
        try:
            c = ctypes.CDLL(ctypes.util.find_library("crypto") or "libcrypto.so.3")
            c.EVP_PKEY_new_raw_public_key.restype = ctypes.c_void_p
            c.EVP_PKEY_new_raw_public_key.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
            c.EVP_MD_CTX_new.restype = ctypes.c_void_p
            c.EVP_MD_CTX_free.argtypes = [ctypes.c_void_p]
            c.EVP_PKEY_free.argtypes = [ctypes.c_void_p]
            c.EVP_DigestVerifyInit.argtypes = [ctypes.c_void_p] * 5
            c.EVP_DigestVerify.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
            c.ERR_clear_error.restype = None
            k = c.EVP_PKEY_new_raw_public_key(1087, None, bytes([1]) + bytes(31), 32)      # EVP_PKEY_ED25519
            if not k:
                raise OSError("no Ed25519 in this libcrypto")
            c.EVP_PKEY_free(k)
            _crypto = c
        except (OSError, AttributeError):
            _crypto = False
    return _crypto or None


def openssl_verify(msg, pk, sig):
    c = libcrypto()
    k = c.EVP_PKEY_new_raw_public_key(1087, None, pk, 32)
    if not k:
        c.ERR_clear_error()
        return False
    ctx = c.EVP_MD_CTX_new()
    ok = c.EVP_DigestVerifyInit(ctx, None, None, None, k) == 1 and c.EVP_DigestVerify(ctx, sig, 64, msg, len(msg)) == 1
    c.EVP_MD_CTX_free(ctx)
    c.EVP_PKEY_free(k)
    c.ERR_clear_error()
    return ok


def encoding_is_canonical(enc, decompress):
    """the 32 bytes are THE encoding of the point they decode to: y < p and no "-0" (x = 0 with the sign bit set).
    `decompress` = oracle/ed25519_ref.decompress (only to find x = 0)."""
    n = int.from_bytes(enc, "little")
    y, sign = n & ((1 << 255) - 1), n >> 255
    if y >= P:
        return False
    p = decompress(enc)
    return not (p is not None and p[0] % P == 0 and sign == 1)


# ---- dalek's POINT laws on libsodium's group arithmetic (round 5) ----------------------------------------------------
# The two verifiers above compare BYTES, dalek's verify_batch compares POINTS - so until round 5 nothing independent pinned
# the labels of the vectors whose R is a non-canonical encoding (y >= p, "-0").  libsodium also exports its group
# primitives: crypto_core_ed25519_add / _sub decode both operands with ge25519_frombytes - y taken mod 2^255 and reduced,
# the sign bit applied to whichever root comes out, "-0" included: dalek's lenient CompressedEdwardsY::decompress - and
# refuse only bytes that are not on the curve.  Scalar multiplication is double-and-add over those two calls alone (the
# library's own crypto_scalarmult_ed25519* refuse small-order, mixed-order and non-canonical points: no use here), the
# hash is hashlib's SHA-512, the scalars are Python integers.  Nothing of oracle/ or of this project's arithmetic is used:
# an independent evaluation of
#     cofactorless   s < L,  A and R decode,      [s]B - [k]A == R        as points    (HS_ED25519_BATCH, per signature)
#     cofactored     the same with [8]([s]B - [k]A - R) == O
#     strict         cofactorless and neither A nor R of small order      (already refereed by crypto_sign_verify_detached)
L_ORDER = 2 ** 252 + 27742317777372353535851937790883648493
_IDENTITY = bytes([1]) + bytes(31)
_BASE = bytes.fromhex("5866666666666666666666666666666666666666666666666666666666666666")


def _sodium_group():
    s = libsodium()
    if s is None or not hasattr(s, "crypto_core_ed25519_add"):
        return None
    for f in (s.crypto_core_ed25519_add, s.crypto_core_ed25519_sub):
        f.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
        f.restype = ctypes.c_int
    return s


def sodium_point_add(p, q, sub=False):
    """canonical encoding of P + Q (P - Q), or None when either operand does not decode to a curve point"""
    s = _sodium_group()
    out = ctypes.create_string_buffer(32)
    rc = (s.crypto_core_ed25519_sub if sub else s.crypto_core_ed25519_add)(out, bytes(p), bytes(q))
    return out.raw if rc == 0 else None


def sodium_point_decode(enc):
    """the canonical encoding of the point `enc` decodes to under the lenient rules, or None (not on the curve)"""
    return sodium_point_add(enc, _IDENTITY)


def sodium_scalar_mult(k, p):
    """[k]P by double-and-add over crypto_core_ed25519_add alone; any k >= 0, any point of the curve (torsion included)"""
    acc, run = _IDENTITY, sodium_point_decode(p)
    if run is None:
        return None
    while k:
        if k & 1:
            acc = sodium_point_add(acc, run)
        run = sodium_point_add(run, run)
        k >>= 1
    return acc


def sodium_point_laws(msg, pk, sig):
    """-> {"cofactorless": bool, "cofactored": bool, "strict": bool} by the definitions above, on libsodium's arithmetic"""
    import hashlib
    r_enc, s = sig[:32], int.from_bytes(sig[32:], "little")
    no = {"cofactorless": False, "cofactored": False, "strict": False}
    if s >= L_ORDER:
        return no
    a, r = sodium_point_decode(pk), sodium_point_decode(r_enc)
    if a is None or r is None:
        return no
    k = int.from_bytes(hashlib.sha512(bytes(r_enc) + bytes(pk) + bytes(msg)).digest(), "little") % L_ORDER
    d = sodium_point_add(sodium_point_add(sodium_scalar_mult(s, _BASE), sodium_scalar_mult(k, a), sub=True), r, sub=True)
    small = lambda p: sodium_scalar_mult(8, p) == _IDENTITY
    cofactorless = d == _IDENTITY
    return {"cofactorless": cofactorless, "cofactored": small(d), "strict": cofactorless and not small(a) and not small(r)}


def sodium_verify_batch(msgs, pks, sigs, zs):
    """ed25519-dalek 1.0.1 `verify_batch` AS IT COMPUTES IT, for given coefficients z_i, on libsodium's group arithmetic:
    every s_i < L and every A_i, R_i decodes (else Err), then  -[sum z_i s_i]B + sum [z_i]R_i + sum [z_i h_i mod L]A_i == O.
    With the z_i fixed the randomised verdict is a deterministic function of the inputs - which makes class (iii)
    (torsion in A or R: the verdict depends on the z_i) refereeable."""
    import hashlib
    acc, bcoef = _IDENTITY, 0
    for msg, pk, sig, z in zip(msgs, pks, sigs, zs):
        s = int.from_bytes(sig[32:], "little")
        if s >= L_ORDER:
            return False
        a, r = sodium_point_decode(pk), sodium_point_decode(sig[:32])
        if a is None or r is None:
            return False
        h = int.from_bytes(hashlib.sha512(bytes(sig[:32]) + bytes(pk) + bytes(msg)).digest(), "little") % L_ORDER
        bcoef = (bcoef + z * s) % L_ORDER
        acc = sodium_point_add(acc, sodium_scalar_mult(z, r))
        acc = sodium_point_add(acc, sodium_scalar_mult(z * h % L_ORDER, a))
    return sodium_point_add(acc, sodium_scalar_mult(bcoef, _BASE), sub=True) == _IDENTITY
