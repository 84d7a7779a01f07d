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
