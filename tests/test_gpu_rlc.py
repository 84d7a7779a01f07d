"""HS_ED25519_BATCH_RLC: dalek's verify_batch as it is computed (one random linear combination
evaluated as a Pippenger MSM on the GPU, hs_rlc.hip) with the per-signature kernels as the
localising fallback.  Parity vs the oracle's own RLC restatement (hs_oracle_verify_batch) and its
per-signature verdicts, at the sizes of SURVEY.md 8c (1, 2, 63, 64, 65, 94, 95 = the reference's
Straus/Pippenger switch, 1,024, 6,667) and on the edge-vector classes where dalek is deterministic."""
import os

import pytest
import torch

import synth
from conftest import bits, oracle_verify_batch, oracle_verify_many
from gpu_helpers import DeviceRunner
from hotstuff_amd import _lib

pytestmark = pytest.mark.gpu
H = bytes.fromhex
CACHE = os.environ.get("HS_SYNTH_CACHE", "/tmp/hs_synth")


@pytest.fixture(scope="module", autouse=True)
def engine():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _lib.init(0, 8192)
    _lib.set_rlc_key(bytes(range(32)))
    yield


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 94, 95, 1024, 6667])
@pytest.mark.parametrize("shared", [True, False])
def test_rlc_accepts_valid_batches_and_localises_bad_ones(oracle, n, shared):
    r = DeviceRunner(max(n, 64), rlc_min=1)
    dg, pk, sg = synth.make_batch(n, shared=shared, cache_dir=CACHE)
    st = 0 if shared else 32
    b = r.bits(dg, pk, sg, n, _lib.BATCH_RLC, shared)
    assert all(b) and oracle_verify_batch(oracle, dg, st, pk, sg, n) == 1
    assert r.ctx.rlc_last()[0] == 1, "the combination itself must have accepted (not the fallback)"
    # corrupt: one bit of one s (keeps every format check green, breaks the equation)
    v = n // 2
    s2 = bytearray(sg); s2[64 * v + 35] ^= 0x10; s2 = bytes(s2)
    expect = oracle_verify_many(oracle, dg, st, pk, s2, n, 0)
    b = r.bits(dg, pk, s2, n, _lib.BATCH_RLC, shared)
    assert b == expect and not all(b)
    assert r.ctx.rlc_last()[0] == 0 and oracle_verify_batch(oracle, dg, st, pk, s2, n) == 0
    if n >= 63:     # the SURVEY.md 8d mix (off-curve points, s >= L, wrong digests ...)
        dg3, pk3, sg3, bad = synth.corrupt(dg, pk, sg, n, shared, rate_byte=20)
        expect = oracle_verify_many(oracle, dg3, st, pk3, sg3, n, 0)
        assert r.bits(dg3, pk3, sg3, n, _lib.BATCH_RLC, shared) == expect
        assert oracle_verify_batch(oracle, dg3, st, pk3, sg3, n) == int(all(expect))
    r.close()


def test_rlc_on_the_deterministic_edge_classes(oracle, edge_vectors):
    """Every edge vector as a one-signature batch.  Class (i) (not even cofactored-valid) must be
    rejected and class (ii) (cofactorless-valid, A and R torsion-free) accepted by the combination
    itself; on the torsion classes dalek's answer depends on z mod 8 - there the call still returns
    one of dalek's two possible answers: the RLC verdict when it accepts, else the per-signature one."""
    r = DeviceRunner(64, rlc_min=1)
    for v in edge_vectors:
        dg, pk, sg = H(v["msg"]), H(v["pk"]), H(v["sig"])
        b = r.bits(dg, pk, sg, 1, _lib.BATCH_RLC, True)
        stage = r.ctx.rlc_last()[0]
        if not v["cofactored"]:
            assert not b[0] and stage == 0, v["name"]
            assert oracle_verify_batch(oracle, dg, 0, pk, sg, 1) == 0
        elif v["cofactorless"] and v["torsion_free"]:
            assert b[0] and stage == 1, v["name"]
            assert oracle_verify_batch(oracle, dg, 0, pk, sg, 1) == 1
        else:
            assert b[0] == (stage == 1 or v["cofactorless"]), v["name"]
    r.close()


def test_rlc_with_the_committee_registered(oracle):
    """A_i served from the key set's Niels records (no decompression of A, 128 B per key)."""
    n = 3000
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
    keys = [pk[32 * i:32 * i + 32] for i in range(n)]
    ks = _lib.KeySet(keys)
    r = DeviceRunner(n, rlc_min=1)
    assert all(r.bits(dg, None, sg, n, _lib.BATCH_RLC, True, keyset=ks, keys=keys)) and r.ctx.rlc_last()[0] == 1
    dg2, pk2, sg2, bad = synth.corrupt(dg, pk, sg, n, True, rate_byte=6)
    keys2 = [pk2[32 * i:32 * i + 32] for i in range(n)]
    expect = bytearray(oracle_verify_many(oracle, dg2, 0, pk2, sg2, n, 0))
    for i in range(n):
        if keys2[i] not in ks.index:
            expect[i] = 0
    assert r.bits(dg2, None, sg2, n, _lib.BATCH_RLC, True, keyset=ks, keys=keys2) == bytes(expect)
    assert r.ctx.rlc_last()[0] == 0
    r.close()
    ks.close()


def test_rlc_through_the_host_api_at_qc_sizes(oracle):
    """hs_ed25519_verify_shared(mode = BATCH_RLC): N = 10,000 and N = 100,000 committees."""
    for n in (6667, 66667):
        dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
        ok, bm = _lib.verify_packed(dg, pk, sg, n, _lib.BATCH_RLC, shared=True, want_bitmap=True)
        assert ok and all(bits(bm, n))
        dg2, pk2, sg2, bad = synth.corrupt(dg, pk, sg, n, True, rate_byte=8)
        ok, bm = _lib.verify_packed(dg2, pk2, sg2, n, _lib.BATCH_RLC, shared=True, want_bitmap=True)
        assert not ok and bits(bm, n) == oracle_verify_many(oracle, dg2, 0, pk2, sg2, n, 0)


def test_small_calls_skip_the_pipeline():
    """Below hs_ed25519_rlc_min_batch() a BATCH_RLC call is plain BATCH: the RLC stage does not run."""
    n = 100
    assert n < _lib.lib().hs_ed25519_rlc_min_batch()
    r = DeviceRunner(n)
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
    assert all(r.bits(dg, pk, sg, n, _lib.BATCH_RLC, True))
    assert r.ctx.rlc_last() == (_lib.RLC_NOT_RUN, 0)
    r.close()


def test_coefficients_are_fresh_per_call_and_per_graph_replay(edge_vectors):
    """dalek draws new coefficients for every verify_batch call, so on a class-(iii) input (torsion in A or R) its verdict
    is an independent chance per call.  Here the key of an execution = process secret ^ host call number ^ a counter on the
    device that every execution bumps - the last one because a round REPLAYED from a captured hipGraph carries the host's
    by-value key unchanged (the ring's GRAPH form gave the same signature the same z_i for ever).  Observable
    through the RLC stage's own verdict on a one-signature batch whose combination holds for one z in eight."""
    import numpy as np
    v = next(v for v in edge_vectors if v["name"].startswith("mixedorder:A k=0mod8"))
    dg, pk, sg = H(v["msg"]), H(v["pk"]), H(v["sig"])
    trials = 128                                      # P(no accept at 1/8 each) = 4e-8
    _lib.set_rlc_key(None)                            # the default: OS secret + counters
    try:
        r = DeviceRunner(64, rlc_min=1)
        stages = []
        for _ in range(trials):
            assert r.bits(dg, pk, sg, 1, _lib.BATCH_RLC, True)[0]             # the call's verdict: per-signature, always accept
            stages.append(r.ctx.rlc_last()[0])
        r.close()
        assert set(stages) == {0, 1}, "device API: the RLC stage gave the same verdict on every call"
        ring = _lib.RoundRing(64, slots=1, kind="graph")
        ctx = ring.slot_ctx(0)
        ctx.set_rlc_min(1)
        stages = []
        for _ in range(trials):
            d, k, s = ring.begin(0, 1, shared=True)
            d[:] = np.frombuffer(dg, np.uint8); k[:] = np.frombuffer(pk, np.uint8); s[:] = np.frombuffer(sg, np.uint8)
            ring.submit(0, _lib.BATCH_RLC)
            ok, bm = ring.wait(0)
            assert ok and bm[0] & 1
            stages.append(ctx.rlc_last()[0])
        captures, launches = ring.stats()
        assert captures == 1 and launches == trials   # one capture, 127 replays
        assert set(stages) == {0, 1}, "graph replays: the RLC stage gave the same verdict on every replay"
        # pinned key: the same verdict every time (reproducible runs), replayed or not.  The pin applies to what is enqueued -
        # or captured - after it, so the ring is built anew (a graph captured before keeps the arguments it was captured with)
        ring.close()
        _lib.set_rlc_key(bytes(range(32)))
        ring = _lib.RoundRing(64, slots=1, kind="graph")
        ctx = ring.slot_ctx(0)
        ctx.set_rlc_min(1)
        stages = []
        for _ in range(16):
            d, k, s = ring.begin(0, 1, shared=True)
            d[:] = np.frombuffer(dg, np.uint8); k[:] = np.frombuffer(pk, np.uint8); s[:] = np.frombuffer(sg, np.uint8)
            ring.submit(0, _lib.BATCH_RLC)
            assert ring.wait(0)[0]
            stages.append(ctx.rlc_last()[0])
        assert len(set(stages)) == 1
        ring.close()
    finally:
        _lib.set_rlc_key(bytes(range(32)))
