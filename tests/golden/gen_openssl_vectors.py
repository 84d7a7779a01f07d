"""Generate tests/golden/openssl_vectors.json with OpenSSL's libcrypto - an
implementation of Ed25519 that is independent of everything in this repo.

Run in the build container (needs libcrypto.so.3 and the `openssl` CLI):
    python tests/golden/gen_openssl_vectors.py

Sections
  rfc8032           RFC 8032 7.1 TEST 1-3 (seed, pk, msg, sig).  The expected
                    values are written below from the RFC; the script asserts
                    that OpenSSL reproduces them before writing the file.
  reference_fixture the reference's own test inputs restated: the 4 key pairs
                    of `StdRng::from_seed([0;32])` (crypto/src/tests/
                    crypto_tests.rs:26-29, consensus/src/tests/common.rs:17-20;
                    rand 0.7 StdRng = ChaCha20, key 0, nonce 0, 32 bytes/key
                    [UPSTREAM-RECALL]), the "Hello, world!" digest of
                    crypto_tests.rs:50-115 and the QC fixture digest of
                    consensus/src/tests/common.rs:129-144, signed by OpenSSL,
                    with the verdict each reference test asserts.
  openssl_sign      64 deterministic (seed, digest) pairs signed by OpenSSL and
                    single-bit tamperings of sig / pk / msg with OpenSSL's own
                    verdict (on these classes OpenSSL and dalek agree).
"""
import ctypes
import hashlib
import json
import os
import struct
import subprocess

crypto = ctypes.CDLL("libcrypto.so.3")
crypto.EVP_PKEY_new_raw_private_key.restype = ctypes.c_void_p
crypto.EVP_PKEY_new_raw_private_key.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
crypto.EVP_PKEY_new_raw_public_key.restype = ctypes.c_void_p
crypto.EVP_PKEY_new_raw_public_key.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
crypto.EVP_PKEY_get_raw_public_key.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]
crypto.EVP_MD_CTX_new.restype = ctypes.c_void_p
crypto.EVP_MD_CTX_free.argtypes = [ctypes.c_void_p]
crypto.EVP_PKEY_free.argtypes = [ctypes.c_void_p]
crypto.EVP_DigestSignInit.argtypes = [ctypes.c_void_p] * 5
crypto.EVP_DigestVerifyInit.argtypes = [ctypes.c_void_p] * 5
crypto.EVP_DigestSign.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p, ctypes.c_size_t]
crypto.EVP_DigestVerify.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
EVP_PKEY_ED25519 = 1087


def ossl_keypair(seed):
    k = crypto.EVP_PKEY_new_raw_private_key(EVP_PKEY_ED25519, None, seed, 32)
    assert k
    out = ctypes.create_string_buffer(32)
    n = ctypes.c_size_t(32)
    assert crypto.EVP_PKEY_get_raw_public_key(k, out, ctypes.byref(n)) == 1
    return k, out.raw


def ossl_sign(seed, msg):
    k, pk = ossl_keypair(seed)
    ctx = crypto.EVP_MD_CTX_new()
    assert crypto.EVP_DigestSignInit(ctx, None, None, None, k) == 1
    sig = ctypes.create_string_buffer(64)
    n = ctypes.c_size_t(64)
    assert crypto.EVP_DigestSign(ctx, sig, ctypes.byref(n), msg, len(msg)) == 1
    crypto.EVP_MD_CTX_free(ctx)
    crypto.EVP_PKEY_free(k)
    return pk, sig.raw


def ossl_verify(pk, msg, sig):
    k = crypto.EVP_PKEY_new_raw_public_key(EVP_PKEY_ED25519, None, pk, 32)
    if not k:
        return False
    ctx = crypto.EVP_MD_CTX_new()
    assert crypto.EVP_DigestVerifyInit(ctx, None, None, None, k) == 1
    r = crypto.EVP_DigestVerify(ctx, sig, 64, msg, len(msg))
    crypto.EVP_MD_CTX_free(ctx)
    crypto.EVP_PKEY_free(k)
    return r == 1


RFC8032 = [  # RFC 8032 section 7.1
    ("9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60",
     "d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a", "",
     "e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e06522490155"
     "5fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b"),
    ("4ccd089b28ff96da9db6c346ec114e0f5b8a319f35aba624da8cf6ed4fb8a6fb",
     "3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c", "72",
     "92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da"
     "085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00"),
    ("c5aa8df43f9f837bedb7442f31dcb7b166d38535076f094b85ce3a2e0b4458f7",
     "fc51cd8e6218a1a38da47ed00230f0580816ed13ba3303ac5deb911548908025", "af82",
     "6291d657deec24024827e69c3abe01a30ce548a284743a445e3680d7db5ac3ac"
     "18ff9b538d16f290ae67f760984dc6594a7c15e9716ed28dc027beceea1ec40a"),
]


def sha512_32(*parts):
    h = hashlib.sha512()
    for p in parts:
        h.update(p)
    return h.digest()[:32]


def main():
    out = {"generator": "tests/golden/gen_openssl_vectors.py",
           "openssl": subprocess.run(["openssl", "version"], capture_output=True, text=True).stdout.strip()}

    rfc = []
    for seed, pk, msg, sig in RFC8032:
        opk, osig = ossl_sign(bytes.fromhex(seed), bytes.fromhex(msg))
        assert opk.hex() == pk and osig.hex() == sig, "OpenSSL does not reproduce RFC 8032 vector"
        rfc.append({"seed": seed, "pk": pk, "msg": msg, "sig": sig})
    out["rfc8032"] = rfc

    # --- reference fixture keys: ChaCha20(key=0, nonce=0) keystream, 32 bytes per key
    ks = subprocess.run(["openssl", "enc", "-chacha20", "-K", "00" * 32, "-iv", "00" * 16],
                        input=bytes(128), capture_output=True).stdout
    assert len(ks) == 128
    keys = []
    for i in range(4):
        seed = ks[32 * i:32 * i + 32]
        _, pk = ossl_keypair(seed)
        keys.append({"seed": seed.hex(), "pk": pk.hex()})
    fx = {"keys": keys}
    # crypto_tests.rs: digest = SHA-512("Hello, world!")[..32]; keys().pop() = key 3, then 2, 1
    hello = sha512_32(b"Hello, world!")
    bad = sha512_32(b"Bad message!")
    pop_order = [3, 2, 1, 0]
    fx["hello_digest"] = hello.hex()
    fx["bad_digest"] = bad.hex()
    fx["hello_sigs"] = [ossl_sign(bytes.fromhex(keys[i]["seed"]), hello)[1].hex() for i in pop_order]
    fx["pop_order"] = pop_order
    # consensus/src/tests/common.rs:129-144: QC{hash: Digest::default(), round: 1}.digest()
    qcd = sha512_32(bytes(32), struct.pack("<Q", 1))
    fx["qc_digest"] = qcd.hex()
    fx["qc_votes"] = [{"pk": keys[i]["pk"], "sig": ossl_sign(bytes.fromhex(keys[i]["seed"]), qcd)[1].hex()}
                      for i in pop_order[:3]]
    fx["expected"] = {
        "verify_valid_signature (crypto_tests.rs:50-61)": True,
        "verify_invalid_signature (crypto_tests.rs:64-77)": False,
        "verify_valid_batch (crypto_tests.rs:80-94)": True,
        "verify_invalid_batch (crypto_tests.rs:97-115)": False,
        "verify_valid_qc (messages_tests.rs:8-10)": True,
    }
    out["reference_fixture"] = fx

    # --- deterministic OpenSSL-signed set + tamperings
    vec = []
    for i in range(64):
        seed = hashlib.sha512(b"hs-key" + struct.pack("<Q", i)).digest()[:32]
        msg = hashlib.sha512(b"hs-msg" + struct.pack("<Q", i)).digest()[:32]
        pk, sig = ossl_sign(seed, msg)
        e = {"seed": seed.hex(), "pk": pk.hex(), "msg": msg.hex(), "sig": sig.hex(), "tamper": []}
        h = hashlib.sha512(b"hs-tamper" + struct.pack("<Q", i)).digest()
        for j, field in enumerate(("sig", "pk", "msg")):
            buf = bytearray({"sig": sig, "pk": pk, "msg": msg}[field])
            pos = h[2 * j] % len(buf)
            bit = h[2 * j + 1] % 8
            buf[pos] ^= 1 << bit
            t = {"sig": sig, "pk": pk, "msg": msg}
            t[field] = bytes(buf)
            ok = ossl_verify(t["pk"], t["msg"], t["sig"])
            e["tamper"].append({"field": field, "byte": pos, "bit": bit, "openssl_accepts": ok})
        vec.append(e)
    out["openssl_sign"] = vec

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "openssl_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
