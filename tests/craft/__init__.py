"""tests/craft - adversarial Ed25519 tuples by construction (libhs_craft.so, craft.c) and the class schedule of the scale
differential (VERDICT r4 "Next round" #3).  Test infrastructure: used by tests/ and tools/differential.py only.

A chunk of n tuples is described by a spec word per tuple (craft.c's bit layout); `schedule(n, rng)` draws the classes:

  honest                 valid signatures (the accept side must stay dense, or a kernel that rejects everything would pass)
  A+T                    A' = [a]B + T_i, i = 1..7: accepts iff [k]T_i = O
  R+T                    R' = [r]B + T_j, j = 1..7: never accepts cofactorless (a cofactored verifier would)
  A+T,R+T                both: accepts iff T_j + [k]T_i = O
  torsion-A              a = 0, A' = T_i (small order): strict rejects; cofactorless accepts iff T_j + [k]T_i = O
  torsion-R              r = 0, R' = T_j alone, honest A
  noncanonical           the same pure-torsion points in their alternative encodings (y + p, "-0") as A', as R', as both
  s+L, s=L-1, L, L+1, 2^252, high bits, 0, s+1, bit flip in s     s outside [0, L) or simply wrong
  byte-flip              one random bit of pk / sig / msg flipped after signing (the round-4 corruption, kept as one class of many)
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libhs_craft.so")

ZERO_A, ZERO_R = 1 << 6, 1 << 7
S_MUT = {"s+L": 1, "s=L-1": 2, "s=L": 3, "s=L+1": 4, "s=2^252": 5, "s|2^253": 6, "s|2^255": 7, "s=0": 8, "s+1": 9, "s^2^127": 10}

CLASSES = ["honest", "A+T", "R+T", "A+T,R+T", "torsion-A", "torsion-A,R+T", "torsion-R", "noncanonical-A", "noncanonical-R",
           "noncanonical-A,R"] + sorted(S_MUT) + ["A+T,s+L", "byte-flip"]
# share of a chunk per class (the rest is honest): the torsion classes get the bulk, they are what the differential is for
WEIGHTS = {"A+T": 0.14, "R+T": 0.07, "A+T,R+T": 0.14, "torsion-A": 0.04, "torsion-A,R+T": 0.08, "torsion-R": 0.03, "noncanonical-A": 0.03,
           "noncanonical-R": 0.03, "noncanonical-A,R": 0.03, "A+T,s+L": 0.01, "byte-flip": 0.03,
           **{k: 0.012 for k in S_MUT}}


def lib():
    if not os.path.exists(LIB):
        raise RuntimeError("tests/craft/libhs_craft.so is missing: run `make -C tests/craft` (or __graft_entry__.build())")
    l = ctypes.CDLL(LIB)
    S, vp = ctypes.c_size_t, ctypes.c_void_p
    l.hs_craft_many.argtypes = [vp, vp, vp, S, vp, vp, ctypes.c_int]
    l.hs_craft_many.restype = None
    l.hs_craft_verify_both_many.argtypes = [vp, vp, vp, S, vp, ctypes.c_int]
    l.hs_craft_verify_both_many.restype = None
    l.hs_craft_is_small_order.argtypes = [vp]
    l.hs_craft_is_small_order.restype = ctypes.c_int
    l.hs_oracle_verify_many.argtypes = [vp, S, S, vp, vp, S, ctypes.c_int, vp, ctypes.c_int]
    l.hs_oracle_verify_many.restype = None
    return l


def schedule(n, rng, pool_seeds):
    """(spec uint32[n], cls uint8[n] index into CLASSES, key_seed_index uint32[n]) for a chunk; pool_seeds = number of distinct
    signing seeds (the key pool the key-cache forms register)."""
    w = np.array([0.0] + [WEIGHTS[c] for c in CLASSES[1:]])
    w[0] = 1.0 - w.sum()
    cls = rng.choice(len(CLASSES), size=n, p=w).astype(np.uint8)
    spec = np.zeros(n, dtype=np.uint32)
    ti = rng.integers(1, 8, size=n, dtype=np.uint32)       # torsion index for A, never 0 where the class says "+T"
    tj = rng.integers(1, 8, size=n, dtype=np.uint32)
    t0 = rng.integers(0, 8, size=n, dtype=np.uint32)       # may be 0: the identity is a torsion point too
    enc = rng.integers(1, 4, size=n, dtype=np.uint32)
    c = lambda name: cls == CLASSES.index(name)
    spec[c("A+T")] = ti[c("A+T")]
    spec[c("R+T")] = tj[c("R+T")] << 3
    m = c("A+T,R+T"); spec[m] = ti[m] | (tj[m] << 3)
    m = c("torsion-A"); spec[m] = t0[m] | ZERO_A
    m = c("torsion-A,R+T"); spec[m] = t0[m] | (tj[m] << 3) | ZERO_A
    m = c("torsion-R"); spec[m] = (t0[m] << 3) | ZERO_R
    # the alternative encodings exist for y in {0, 1, p - 1}: T_0 (identity), T_4 (order 2), T_2 / T_6 (order 4)
    even = (rng.integers(0, 4, size=n, dtype=np.uint32) * 2).astype(np.uint32)
    even2 = (rng.integers(0, 4, size=n, dtype=np.uint32) * 2).astype(np.uint32)
    m = c("noncanonical-A"); spec[m] = even[m] | ZERO_A | (enc[m] << 8) | (t0[m] << 3)
    m = c("noncanonical-R"); spec[m] = (even[m] << 3) | ZERO_R | (enc[m] << 10)
    m = c("noncanonical-A,R"); spec[m] = even[m] | (even2[m] << 3) | ZERO_A | ZERO_R | (enc[m] << 8) | (((enc[m] * 2) % 3 + 1) << 10)
    for name, code in S_MUT.items():
        spec[c(name)] = code << 12
    m = c("A+T,s+L"); spec[m] = ti[m] | (1 << 12)
    key = rng.integers(0, pool_seeds, size=n, dtype=np.uint32)
    return spec, cls, key


def key_pool(pool_seeds, base_seed=7):
    """pool_seeds x 32 deterministic signing seeds"""
    rng = np.random.default_rng(base_seed)
    return rng.integers(0, 256, size=(pool_seeds, 32), dtype=np.uint8)


def craft_chunk(l, n, rng, seeds_pool, threads):
    """one chunk: dict of numpy arrays msgs (n, 32), pk (n, 32), sig (n, 64), spec, cls, expect (n,) with bit 0 = BATCH verdict,
    bit 1 = STRICT verdict (the oracle's labels)."""
    spec, cls, key = schedule(n, rng, len(seeds_pool))
    seeds = np.ascontiguousarray(seeds_pool[key])
    msgs = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    pk = np.zeros((n, 32), dtype=np.uint8)
    sig = np.zeros((n, 64), dtype=np.uint8)
    l.hs_craft_many(seeds.ctypes.data, msgs.ctypes.data, spec.ctypes.data, n, pk.ctypes.data, sig.ctypes.data, threads)
    # byte-flip: one bit of pk / sig / msg after signing
    m = np.nonzero(cls == CLASSES.index("byte-flip"))[0]
    if len(m):
        where = rng.integers(0, 128 * 8, size=len(m))
        for i, w in zip(m, where):
            byte, bit = int(w) // 8, int(w) % 8
            arr, off = (pk, byte) if byte < 32 else ((sig, byte - 32) if byte < 96 else (msgs, byte - 96))
            arr[i, off] ^= 1 << bit
    expect = np.zeros(n, dtype=np.uint8)
    l.hs_craft_verify_both_many(msgs.ctypes.data, pk.ctypes.data, sig.ctypes.data, n, expect.ctypes.data, threads)
    return {"msgs": msgs, "pk": pk, "sig": sig, "spec": spec, "cls": cls, "key": key, "expect": expect}


def check_labels_against_the_oracle(l, chunk, rng, sample, threads):
    """verify_both (one pass) against the oracle's two entry points on a random sample of the chunk; returns mismatches"""
    n = len(chunk["cls"])
    idx = np.sort(rng.choice(n, size=min(sample, n), replace=False))
    msgs, pk, sig = (np.ascontiguousarray(chunk[k][idx]) for k in ("msgs", "pk", "sig"))
    out0 = np.zeros(len(idx), dtype=np.uint8)
    out1 = np.zeros(len(idx), dtype=np.uint8)
    l.hs_oracle_verify_many(msgs.ctypes.data, 32, 32, pk.ctypes.data, sig.ctypes.data, len(idx), 0, out0.ctypes.data, threads)
    l.hs_oracle_verify_many(msgs.ctypes.data, 32, 32, pk.ctypes.data, sig.ctypes.data, len(idx), 1, out1.ctypes.data, threads)
    got = chunk["expect"][idx]
    return int(np.count_nonzero((got & 1) != out0) + np.count_nonzero(((got >> 1) & 1) != out1))
