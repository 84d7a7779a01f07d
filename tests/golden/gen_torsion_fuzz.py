"""Build tests/golden/torsion_fuzz.json: randomized torsion-mix signatures labelled by the
big-int ground truth (oracle/ed25519_ref.py).  Complements edge_vectors.json, which has one
or two hand-built vectors per class.

For each case: honest key a (A0 = aB), torsion points Ta, Tr drawn from the 8-torsion subgroup
(possibly the identity), A = A0 + Ta, R = rB + Tr, s = r + k a with k = H(R || A || M).
Then sB - kA - R = -k Ta - Tr: cofactorless-valid iff that torsion point is the identity,
always cofactored-valid.  Some cases additionally re-encode non-canonically where possible,
flip the sign bit, or perturb s by +-L multiples / bit flips.

    python tests/golden/gen_torsion_fuzz.py   (deterministic; ~1 minute)
"""
import hashlib
import json
import os
import struct
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "oracle"))
import ed25519_ref as E  # noqa: E402


def H(tag, i):
    return hashlib.sha512(tag + struct.pack("<Q", i)).digest()


def main(count=160):
    tors = E.eight_torsion()
    out = []
    for i in range(count):
        h = H(b"fuzz", i)
        seed, msg = h[:32], H(b"fuzz-msg", i)[:32]
        a, prefix = E.secret_expand(seed)
        a %= E.L
        Ta, Tr = tors[h[32] % 8], tors[h[33] % 8]
        if h[34] % 4 == 0:
            Ta = E.IDENT
        if h[35] % 4 == 0:
            Tr = E.IDENT
        A = E.compress(E.pt_add(E.pt_mul(a, E.B), Ta))
        r = E.sha512_modl(prefix, msg)
        R = E.compress(E.pt_add(E.pt_mul(r, E.B), Tr))
        k = E.sha512_modl(R, A, msg)
        s = (r + k * a) % E.L
        kind = "torsion-mix"
        v = h[36] % 8
        if v == 1:                      # flip R's sign bit
            R = R[:31] + bytes([R[31] ^ 0x80]); kind = "R-sign-flipped"
        elif v == 2:                    # flip A's sign bit (changes k as well: A bytes are hashed)
            A = A[:31] + bytes([A[31] ^ 0x80]); kind = "A-sign-flipped"
        elif v == 3:                    # s + L  (non-canonical scalar)
            s = s + E.L; kind = "s-plus-L"
        elif v == 4:                    # one bit of s flipped
            s ^= 1 << (h[37] % 252); kind = "s-bit-flip"
        sig = R + int(s % (1 << 256)).to_bytes(32, "little")
        out.append({"kind": kind, "msg": msg.hex(), "pk": A.hex(), "sig": sig.hex(),
                    "strict": E.verify_strict(msg, A, sig),
                    "cofactorless": E.verify_cofactorless(msg, A, sig),
                    "cofactored": E.verify_cofactored(msg, A, sig)})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "torsion_fuzz.json")
    with open(path, "w") as f:
        json.dump({"generator": "tests/golden/gen_torsion_fuzz.py", "vectors": out}, f, indent=0)
    n = len(out)
    print("wrote", path, n, "vectors;",
          "strict-accept", sum(v["strict"] for v in out),
          "cofactorless-accept", sum(v["cofactorless"] for v in out),
          "cofactored-only", sum(v["cofactored"] and not v["cofactorless"] for v in out))


if __name__ == "__main__":
    main()
