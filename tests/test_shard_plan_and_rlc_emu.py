"""CPU-only: (1) the shard plan of the multi-GPU entry points (hs_ed25519_shard_plan, pure host
arithmetic behind the C ABI) - ragged and empty shards, 64-signature alignment, exact cover;
(2) the batch random-linear-combination check (hotstuff_amd/csrc/rlc.h), every stage run on the CPU
by tests/emu against the oracle's restatement of dalek's verify_batch."""
import ctypes
import hashlib
import os
import struct

import pytest

import synth
from conftest import S, oracle_verify_batch, oracle_verify_many
from hotstuff_amd import _lib

H = bytes.fromhex
CACHE = os.environ.get("HS_SYNTH_CACHE", "/tmp/hs_synth")
KEY = bytes(range(32))


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 200, 1000, 6667, 66667, 1 << 20])
@pytest.mark.parametrize("k", [1, 2, 3, 4, 8])
def test_shard_plan_covers_exactly(n, k):
    spans = [_lib.shard_plan(n, k, s) for s in range(k)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    for (lo, hi), (lo2, hi2) in zip(spans, spans[1:]):
        assert hi == lo2 and lo <= hi
    for lo, hi in spans:
        assert lo % 64 == 0 or lo == n                  # slices are whole 64-bit ballots
    sizes = [hi - lo for lo, hi in spans]
    assert all(a >= b for a, b in zip(sizes, sizes[1:]))       # full shards first, then the ragged one, then empty ones
    # same cut as the torch.distributed path (hotstuff_amd/sharding.py)
    from hotstuff_amd.sharding import shard_bounds
    for s in range(k):
        lo, hi, _ = shard_bounds(n, k, s)
        assert (lo, hi) == spans[s]
    # BASELINE config 4: 66,667 signatures on 8 GPUs -> 7 x 8,384 + 7,979
    if (n, k) == (66667, 8):
        assert sizes == [8384] * 7 + [7979]
    if (n, k) == (200, 3):
        assert sizes == [128, 72, 0]                    # ragged + empty tail shard


def test_shard_plan_rejects_bad_arguments():
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    assert _lib.lib().hs_ed25519_shard_plan(10, 0, 0, ctypes.byref(lo), ctypes.byref(hi)) == -3
    assert _lib.lib().hs_ed25519_shard_plan(10, 2, 2, ctypes.byref(lo), ctypes.byref(hi)) == -3


def test_rlc_coefficient_is_keyed_sha512(emu):
    k, s = hashlib.sha512(b"k").digest()[:32], hashlib.sha512(b"s").digest()[:32]
    for index in (0, 1, 66666):
        z = ctypes.create_string_buffer(16)
        emu.emu_rlc_coeff(KEY, k, s, index, z)
        want = bytearray(hashlib.sha512(KEY + k + s + struct.pack("<Q", index)).digest()[:16])
        want[15] &= 0x7f                                  # 127 bits: no carry window in the signed recoding
        assert z.raw == bytes(want)


@pytest.mark.parametrize("n,shared,segments", [(1, True, 1), (2, False, 2), (3, True, 1), (64, False, 3), (95, True, 2), (300, True, 4)])
def test_rlc_emulation_matches_oracle_batch(emu, oracle, n, shared, segments):
    dg, pk, sg = synth.make_batch(n, shared=shared, cache_dir=CACHE)
    st = 0 if shared else 32
    assert emu.emu_rlc_verify(dg, S(st), pk, sg, S(n), KEY, segments) == 1 == oracle_verify_batch(oracle, dg, st, pk, sg, n)
    # one flipped bit of one s: all format checks pass, the combination must not be the identity
    s2 = bytearray(sg); s2[64 * (n // 2) + 33] ^= 4; s2 = bytes(s2)
    assert emu.emu_rlc_verify(dg, S(st), pk, s2, S(n), KEY, segments) == 0 == oracle_verify_batch(oracle, dg, st, pk, s2, n)
    # a wrong digest (per-signature digests only)
    if not shared:
        d2 = bytearray(dg); d2[32 * (n - 1) + 7] ^= 1; d2 = bytes(d2)
        assert emu.emu_rlc_verify(d2, S(st), pk, sg, S(n), KEY, segments) == 0


def test_rlc_emulation_on_deterministic_edge_classes(emu, oracle, edge_vectors):
    """class (i): not cofactored-valid -> reject; class (ii): cofactorless-valid and torsion-free -> accept
    (SURVEY.md 8a A4-note as refined in DESIGN.md section 3)"""
    seen = [0, 0]
    for v in edge_vectors:
        dg, pk, sg = H(v["msg"]), H(v["pk"]), H(v["sig"])
        got = emu.emu_rlc_verify(dg, S(0), pk, sg, S(1), KEY, 1)
        if not v["cofactored"]:
            assert got == 0 == oracle_verify_batch(oracle, dg, 0, pk, sg, 1), v["name"]
            seen[0] += 1
        elif v["cofactorless"] and v["torsion_free"]:
            assert got == 1 == oracle_verify_batch(oracle, dg, 0, pk, sg, 1), v["name"]
            seen[1] += 1
    assert seen[0] >= 10 and seen[1] >= 3


@pytest.mark.parametrize("n,shards", [(1, 3), (65, 2), (200, 3), (300, 8), (129, 2)])
def test_one_combination_cut_over_shards_equals_the_whole(emu, oracle, n, shards):
    """SURVEY.md 8e "batch-MSM mode" on the CPU (the functions hs_rlc.hip / hs_shard.hip run): every shard reduces its
    signatures to a partial record under the call's key - coefficients drawn for positions in the WHOLE call - and the records
    are joined and tested once.  Same verdict as the unsharded combination and as the oracle's restatement of dalek's
    verify_batch, for valid batches, a broken equation, and a malformed signature (a shard's `failed` word reaching the join)."""
    for shared in (True, False):
        dg, pk, sg = synth.make_batch(n, shared=shared, cache_dir=CACHE)
        st = 0 if shared else 32
        assert emu.emu_rlc_verify_sharded(dg, S(st), pk, sg, S(n), KEY, shards) == 1 == oracle_verify_batch(oracle, dg, st, pk, sg, n)
        s2 = bytearray(sg); s2[64 * (n - 1) + 33] ^= 4; s2 = bytes(s2)          # the LAST shard's last signature
        assert emu.emu_rlc_verify_sharded(dg, S(st), pk, s2, S(n), KEY, shards) == 0 == oracle_verify_batch(oracle, dg, st, pk, s2, n)
        s3 = bytearray(sg); s3[64 * 0 + 63] |= 0x10; s3 = bytes(s3)             # s >= L in the first shard
        assert emu.emu_rlc_verify_sharded(dg, S(st), pk, s3, S(n), KEY, shards) == 0 == oracle_verify_batch(oracle, dg, st, pk, s3, n)
    if n == 200:      # two signatures that would cancel if they shared a coefficient: the same tuple with s negated cannot be built, but
        # the same tuple twice in different shards must get different coefficients - swapped halves still verify
        dg, pk, sg = synth.make_batch(100, shared=True, cache_dir=CACHE)
        assert emu.emu_rlc_verify_sharded(dg, S(0), pk + pk, sg + sg, S(200), KEY, shards) == 1
