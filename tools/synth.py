"""Deterministic synthetic (digest, pk, sig) workloads.

  key seed i      SHA-512("hs-key" || le64(i))[..32]
  shared digest   SHA-512("hs-qc"  || le64(round))[..32]   (QC::verify shape)
  digest i        SHA-512("hs-msg" || le64(i))[..32]        (TC::verify / multi shape)
  signature i     RFC 8032 deterministic signature

Signing is the reference's CPU-side ``Signature::new`` (crypto/src/lib.rs:185-191) and is
out of scope for the GPU engine; this tool signs with the system OpenSSL (libcrypto),
which is independent of both the engine and oracle/.  Used by tests/ and bench.py.
"""
import ctypes
import ctypes.util
import hashlib
import os
import struct

import numpy as np

_crypto = None


def _libcrypto():
    global _crypto
    if _crypto is None:
        name = ctypes.util.find_library("crypto") or "libcrypto.so.3"
        c = ctypes.CDLL(name)
        c.EVP_PKEY_new_raw_private_key.restype = ctypes.c_void_p
        c.EVP_PKEY_new_raw_private_key.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        c.EVP_PKEY_get_raw_public_key.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]
        c.EVP_MD_CTX_new.restype = ctypes.c_void_p
        c.EVP_MD_CTX_free.argtypes = [ctypes.c_void_p]
        c.EVP_PKEY_free.argtypes = [ctypes.c_void_p]
        c.EVP_DigestSignInit.argtypes = [ctypes.c_void_p] * 5
        c.EVP_DigestSign.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t),
                                     ctypes.c_char_p, ctypes.c_size_t]
        _crypto = c
    return _crypto


def key_seed(i):
    return hashlib.sha512(b"hs-key" + struct.pack("<Q", i)).digest()[:32]


def shared_digest(round_=1):
    return hashlib.sha512(b"hs-qc" + struct.pack("<Q", round_)).digest()[:32]


def msg_digest(i):
    return hashlib.sha512(b"hs-msg" + struct.pack("<Q", i)).digest()[:32]


def sign(seed, msg):
    """(pk, sig) by OpenSSL."""
    c = _libcrypto()
    k = c.EVP_PKEY_new_raw_private_key(1087, None, seed, 32)
    if not k:
        raise RuntimeError("OpenSSL: cannot load Ed25519 key")
    pk = ctypes.create_string_buffer(32)
    n = ctypes.c_size_t(32)
    c.EVP_PKEY_get_raw_public_key(k, pk, ctypes.byref(n))
    ctx = c.EVP_MD_CTX_new()
    c.EVP_DigestSignInit(ctx, None, None, None, k)
    sig = ctypes.create_string_buffer(64)
    n = ctypes.c_size_t(64)
    ok = c.EVP_DigestSign(ctx, sig, ctypes.byref(n), msg, len(msg))
    c.EVP_MD_CTX_free(ctx)
    c.EVP_PKEY_free(k)
    if ok != 1:
        raise RuntimeError("OpenSSL: EVP_DigestSign failed")
    return pk.raw, sig.raw


def openssl_verify_rate(digests, pks, sigs, n, stride, budget_s=0.5):
    """EVP_DigestVerify in a loop on one core over the first tuples of a packed batch (all must verify):
    the sanity bracket of SURVEY.md 8d for single verifications.  Returns verifications per second.
    (The ctypes call overhead, ~1 us per call, is inside the figure; a verification is ~50-100 us.)"""
    import time
    c = _libcrypto()
    c.EVP_PKEY_new_raw_public_key.restype = ctypes.c_void_p
    c.EVP_PKEY_new_raw_public_key.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    c.EVP_DigestVerifyInit.argtypes = [ctypes.c_void_p] * 5
    c.EVP_DigestVerify.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    done, i = 0, 0
    t0 = time.perf_counter()
    while True:
        k = c.EVP_PKEY_new_raw_public_key(1087, None, pks[32 * i:32 * i + 32], 32)
        ctx = c.EVP_MD_CTX_new()
        c.EVP_DigestVerifyInit(ctx, None, None, None, k)
        d = digests[32 * i:32 * i + 32] if stride else digests[:32]
        ok = c.EVP_DigestVerify(ctx, sigs[64 * i:64 * i + 64], 64, d, 32)
        c.EVP_MD_CTX_free(ctx)
        c.EVP_PKEY_free(k)
        if ok != 1:
            raise RuntimeError("OpenSSL rejected synthetic signature %d" % i)
        done += 1
        i = (i + 1) % n
        if done % 64 == 0 and time.perf_counter() - t0 > budget_s:
            break
    return done / (time.perf_counter() - t0)


def sodium_verify_rate(digests, pks, sigs, n, stride, budget_s=0.5):
    """libsodium's crypto_sign_verify_detached in a loop on one core: the second bracket for single verifications (a widely
    deployed ref10-lineage implementation, the same lineage as dalek's serial backend).  Verifications per second; raises
    when the image has no libsodium."""
    import ctypes.util
    import glob
    import time
    lib = None
    for c in [os.environ.get("HS_LIBSODIUM"), ctypes.util.find_library("sodium")] + sorted(glob.glob("/opt/conda/lib/libsodium.so*")) + \
            sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libsodium.so*")):
        if not c:
            continue
        try:
            lib = ctypes.CDLL(c)
            if lib.sodium_init() >= 0:
                break
            lib = None
        except OSError:
            lib = None
    if lib is None:
        raise RuntimeError("no libsodium in this image")
    lib.crypto_sign_verify_detached.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_ulonglong, ctypes.c_char_p]
    done, i = 0, 0
    t0 = time.perf_counter()
    while True:
        d = digests[32 * i:32 * i + 32] if stride else digests[:32]
        if lib.crypto_sign_verify_detached(sigs[64 * i:64 * i + 64], d, 32, pks[32 * i:32 * i + 32]) != 0:
            raise RuntimeError("libsodium rejected synthetic signature %d" % i)
        done += 1
        i = (i + 1) % n
        if done % 64 == 0 and time.perf_counter() - t0 > budget_s:
            break
    return done / (time.perf_counter() - t0)


def make_batch(n, shared, round_=1, first=0, cache_dir=None):
    """Returns (digests, pks, sigs) as packed bytes: digests is 32 bytes when `shared`
    else 32*n.  All signatures valid.  Cached under cache_dir when given."""
    tag = "hs_synth_%s_%d_%d_%d.npz" % ("shared" if shared else "multi", n, round_, first)
    path = os.path.join(cache_dir, tag) if cache_dir else None
    if path and os.path.exists(path):
        z = np.load(path)
        return z["d"].tobytes(), z["p"].tobytes(), z["s"].tobytes()
    d_sh = shared_digest(round_)
    pks, sigs, dgs = bytearray(), bytearray(), bytearray()
    for i in range(first, first + n):
        m = d_sh if shared else msg_digest(i)
        pk, sg = sign(key_seed(i), m)
        pks += pk
        sigs += sg
        if not shared:
            dgs += m
    digests = d_sh if shared else bytes(dgs)
    if path:
        os.makedirs(cache_dir, exist_ok=True)
        np.savez(path, d=np.frombuffer(digests, np.uint8), p=np.frombuffer(bytes(pks), np.uint8),
                 s=np.frombuffer(bytes(sigs), np.uint8))
    return digests, bytes(pks), bytes(sigs)


def corrupt(digests, pks, sigs, n, shared, rate_byte=8):
    """The parity mix of SURVEY.md 8d: signature i is corrupted iff
    SHA-512("hs-bad" || le64(i))[0] < rate_byte; the corruption class is byte 1 mod 4:
    0 flip a bit of R, 1 flip a bit of s, 2 flip a bit of pk, 3 flip a bit of the digest
    (per-signature digests only; falls back to class 0 for a shared digest).
    Returns (digests, pks, sigs, bad_indices)."""
    digests, pks, sigs = bytearray(digests), bytearray(pks), bytearray(sigs)
    bad = []
    for i in range(n):
        h = hashlib.sha512(b"hs-bad" + struct.pack("<Q", i)).digest()
        if h[0] >= rate_byte:
            continue
        cls = h[1] % 4
        pos, bit = h[2] % 32, h[3] % 8
        if cls == 3 and shared:
            cls = 0
        if cls == 0:
            sigs[64 * i + pos] ^= 1 << bit
        elif cls == 1:
            sigs[64 * i + 32 + pos] ^= 1 << bit
        elif cls == 2:
            pks[32 * i + pos] ^= 1 << bit
        else:
            digests[32 * i + pos] ^= 1 << bit
        bad.append(i)
    return bytes(digests), bytes(pks), bytes(sigs), bad
