"""Bit-exact GPU-vs-C-oracle parity AT BASELINE.json's sizes, ~3 % of the signatures corrupted
(SURVEY.md 8d mix): n = 6,667 (config 3: N = 10,000), n = 66,667 (config 4: N = 100,000; cold
two-kernel path and the cached three-wavefront kernel, by size), n = 8,384 (the shard one of 8 GPUs
gets) and n > 98,304 (the one-lane cached kernel reached by size, not by a selector).  Cold and
key-cache paths, both modes, through the host-pointer C ABI."""
import os

import pytest
import torch

import synth
from conftest import bits, oracle_verify_many
from hotstuff_amd import _lib

pytestmark = pytest.mark.gpu
CACHE = os.environ.get("HS_SYNTH_CACHE", "/tmp/hs_synth")


@pytest.fixture(scope="module", autouse=True)
def engine():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _lib.init(0, 8192)
    yield


def _mixed(n):
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
    dg, pk, sg, bad = synth.corrupt(dg, pk, sg, n, True, rate_byte=8)       # ~3 %
    assert bad and len(bad) < n // 10
    return dg, pk, sg, bad


@pytest.mark.parametrize("n", [6667, 8384, 20000, 66667])      # eight lanes / four lanes 4-bit / four lanes 3-bit / one-lane pair with the sliver in its stage-2 grid
def test_cold_and_cached_vs_oracle_at_baseline_size(oracle, n):
    dg, pk, sg, bad = _mixed(n)
    keys = [pk[32 * i:32 * i + 32] for i in range(n)]
    ks = _lib.KeySet(keys)                       # 66,667 keys: 44 GB of tables
    packed, idx = ks.indices(keys)
    for mode in (_lib.BATCH, _lib.STRICT):
        expect = oracle_verify_many(oracle, dg, 0, pk, sg, n, mode)
        assert 0 < sum(1 for b in expect if not b) <= len(bad)
        ok, bm = _lib.verify_packed(dg, pk, sg, n, mode, shared=True, want_bitmap=True)
        assert not ok and bits(bm, n) == expect, "cold path, mode %d" % mode
        ok, bm = ks.verify_packed(dg, packed, sg, n, mode, shared=True, want_bitmap=True)
        assert not ok and bits(bm, n) == expect, "key-cache path, mode %d" % mode
    # 4,097 ... 8,192 signatures: whichever of the two eight-lane forms hs_ed25519_autotune measured faster on this GPU at init
    assert _lib.lib().hs_ed25519_dominant_kernel_for(n).decode() in (("k_verify_fused8h", "k_verify_fused8") if n <= 8192 else (("k_verify_fused" if n <= 16384 else ("k_verify_fused3" if n <= 32768 else "k_verify_tail")),))
    assert _lib.lib().hs_ed25519_cached_kernel_for(n).decode() == "k_verify_cached_lane2r"
    ks.close()


@pytest.mark.parametrize("n", [6667, 8384, 66667])
def test_pairs_form_vs_oracle_at_baseline_size(oracle, n):
    """The pairs form of the fused kernel (two signatures per quad, 32 per wavefront) forced on one device-resident
    batch of BASELINE size, ~3 % corrupted, both modes, shared and per-signature digests."""
    from gpu_helpers import DeviceRunner
    run = DeviceRunner(n, kernel=_lib.KERNEL_FUSED_PAIRS)
    for shared in (True, False):
        dg, pk, sg = synth.make_batch(n, shared=shared, cache_dir=CACHE)
        dg, pk, sg, bad = synth.corrupt(dg, pk, sg, n, shared, rate_byte=8)
        for mode in (_lib.BATCH, _lib.STRICT):
            expect = oracle_verify_many(oracle, dg, 0 if shared else 32, pk, sg, n, mode)
            assert run.bits(dg, pk, sg, n, mode, shared) == expect
    run.close()


@pytest.mark.parametrize("reps,kernel", [(193, ("k_cached_chain2",)), (98, ("k_cached_chain2",))])      # 197,632 and 100,352
def test_large_cached_call_takes_the_batched_finish_by_size(oracle, reps, kernel):
    """n = 197,632: hs_ed25519_verify_shared_cached dispatches the chains + batched-finish pair by size (a lone call: four
    signatures per finish lane), from 98,305 signatures on every GPU (round 5 retired the per-box switch between 98,305 and 196,607).  1,024 distinct committee keys, their votes tiled, ~3 % corrupted
    individually."""
    base = 1024
    dg, pk, sg = synth.make_batch(base, shared=True, cache_dir=CACHE)
    n = base * reps
    assert n > 98304 and _lib.lib().hs_ed25519_cached_kernel_for(n).decode() in kernel
    pk_t, sg_t = pk * reps, sg * reps
    dg2, pk_t, sg_t, bad = synth.corrupt(dg, pk_t, sg_t, n, True, rate_byte=8)
    keys = [pk[32 * i:32 * i + 32] for i in range(base)]
    ks = _lib.KeySet(keys)
    import struct
    # a corrupted key byte makes the key unknown to the set: index outside it, rejected like the cold path rejects it
    idx = [ks.index.get(pk_t[32 * i:32 * i + 32], _lib.KeySet.UNKNOWN) for i in range(n)]
    packed = struct.pack("<%dI" % n, *idx)
    for mode in (_lib.BATCH, _lib.STRICT):
        expect = bytearray(oracle_verify_many(oracle, dg2, 0, pk_t, sg_t, n, mode))
        for i in range(n):
            if idx[i] == _lib.KeySet.UNKNOWN:
                expect[i] = 0                    # not a committee member (the oracle may accept a flipped key bit by chance: never here)
        ok, bm = ks.verify_packed(dg2, packed, sg_t, n, mode, shared=True, want_bitmap=True)
        assert not ok and bits(bm, n) == bytes(expect)
    ks.close()


@pytest.mark.parametrize("n", [6667, 66667])
def test_concurrent_hint_changes_the_form_not_the_verdicts(oracle, n):
    """hs_ed25519_ctx_set_concurrent: four lanes per signature instead of eight at 6,667, no side-stream sliver at 66,667 -
    and the same accept bits as the oracle, both modes, RLC stage in front included (which never splits the sliver off)."""
    from gpu_helpers import DeviceRunner
    dg, pk, sg, bad = _mixed(n)
    runs = [DeviceRunner(n), DeviceRunner(n), DeviceRunner(n, rlc_min=1)]
    assert runs[1].ctx.set_concurrent(True) == 0 and runs[1].ctx.set_concurrent(True) == 1      # returns the previous setting
    for mode in (_lib.BATCH, _lib.STRICT):
        expect = oracle_verify_many(oracle, dg, 0, pk, sg, n, mode)
        for r in runs[:2]:
            assert r.bits(dg, pk, sg, n, mode, True) == expect
    expect = oracle_verify_many(oracle, dg, 0, pk, sg, n, _lib.BATCH)
    assert runs[2].bits(dg, pk, sg, n, _lib.BATCH_RLC, True) == expect                         # rejected by the RLC, localised behind it
    vdg, vpk, vsg = synth.make_batch(n, shared=True, cache_dir=CACHE)
    assert all(runs[2].bits(vdg, vpk, vsg, n, _lib.BATCH_RLC, True)) and runs[2].ctx.rlc_last()[0] == 1
    if n >= 49152:
        # the key cache follows the hint too: a context that is not alone takes two-lane chains + the batched finish by compression
        # from 49,152 signatures (k_cached_chain2 + k_cached_finish) where a lone one keeps the R-wavefront kernel - same bits
        keys = [pk[32 * i:32 * i + 32] for i in range(n)]
        ks = _lib.KeySet(keys)
        for mode in (_lib.BATCH, _lib.STRICT):
            expect = oracle_verify_many(oracle, dg, 0, pk, sg, n, mode)
            for r in runs[:2]:
                assert r.bits(dg, None, sg, n, mode, True, keyset=ks, keys=keys) == expect
        ks.close()
    for r in runs:
        r.close()


@pytest.mark.parametrize("n,kernel", [(65536 + 16384, "k_verify_tail"), (65536 + 16448, "k_verify"), (2 * 65536 + 100, "k_verify_tail")])
def test_sliver_past_whole_rounds_of_wavefronts(oracle, n, kernel):
    """Large one-lane calls: a sliver of up to 16,384 signatures past whole rounds of wavefronts rides in the stage-2 launch
    as four-lane workgroups (k_verify_tail); a larger one does not.  Cold path, both modes, ~3 % corrupted, vs the oracle."""
    assert _lib.lib().hs_ed25519_dominant_kernel_for(n).decode() == kernel
    dg, pk, sg, bad = _mixed(n)
    for mode in (_lib.BATCH, _lib.STRICT):
        expect = oracle_verify_many(oracle, dg, 0, pk, sg, n, mode)
        ok, bm = _lib.verify_packed(dg, pk, sg, n, mode, shared=True, want_bitmap=True)
        assert not ok and bits(bm, n) == expect, "mode %d" % mode


def test_one_call_of_a_million_signatures(oracle):
    """Maximum-size edge: ONE host call of 1,048,653 signatures (ragged: not a multiple of 64; 16 times config 4's QC) - the
    bulk-copy staging, a 2.7 GB workspace, whole rounds of one-lane wavefronts plus a sliver - ~0.4 % corrupted individually,
    bit-exact against the oracle in BATCH and STRICT, and in BATCH_RLC (the combination rejects, the per-signature kernels
    localise: BATCH's bitmap); the uncorrupted set is accepted by the combination itself."""
    base, n = 1024, 1024 * 1024 + 77
    dg, pk, sg = synth.make_batch(base, shared=True, cache_dir=CACHE)
    reps = (n + base - 1) // base
    pk_t, sg_t = (pk * reps)[:32 * n], (sg * reps)[:64 * n]
    dg2, pk_c, sg_c, bad = synth.corrupt(dg, pk_t, sg_t, n, True, rate_byte=1)
    assert 1000 < len(bad) < n // 100
    for mode in (_lib.BATCH, _lib.STRICT):
        expect = oracle_verify_many(oracle, dg2, 0, pk_c, sg_c, n, mode)
        assert 0 < sum(1 for b in expect if not b) <= len(bad)
        ok, bm = _lib.verify_packed(dg2, pk_c, sg_c, n, mode, shared=True, want_bitmap=True)
        assert not ok and bits(bm, n) == expect, "mode %d" % mode
        if mode == _lib.BATCH:
            ok, bm = _lib.verify_packed(dg2, pk_c, sg_c, n, _lib.BATCH_RLC, shared=True, want_bitmap=True)
            assert not ok and bits(bm, n) == expect, "BATCH_RLC localisation"
    ok, bm = _lib.verify_packed(dg, pk_t, sg_t, n, _lib.BATCH_RLC, shared=True, want_bitmap=True)
    assert ok and all(bits(bm, n))
    ok, _ = _lib.verify_packed(dg, pk_t, sg_t, n, _lib.STRICT, shared=True)
    assert ok
