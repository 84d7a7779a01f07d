"""Round-2 sweeps on one MI355X (python tools/r2_sweeps.py > profiles/r2_sweeps.md):
  1. cold QC path, per-signature BATCH kernels vs HS_ED25519_BATCH_RLC (one random linear combination = Pippenger
     MSM, per-signature fallback skipped when it accepts): p50 of one device-resident call and signatures/s with four
     contexts in flight, n = 2,048 ... 262,144 -> where the library's threshold belongs
  2. per-signature cold path above the fused kernel's default range: k_verify_fused (four lanes per signature,
     table in LDS) forced at every size vs k_prepare + k_verify (one lane per signature, table in private memory)
  3. key-cache path: k_verify_cached_lane2r (two lanes per signature + an R-decompression wavefront) vs the two-launch form
     (k_cached_chain2 + k_cached_finish) by size.  (profiles/r2_sweeps.md section 3 compares it with the one-lane kernel
     k_verify_cached_lane, which left the library in round 5.)
Synthetic sets are signed on the GPU (hs_ed25519_public_keys / hs_ed25519_sign_batch)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import synth
from hotstuff_amd import _lib

_lib.init(0, 8192)
dev = torch.device("cuda", 0)
t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)


def make(n, shared=True):
    seeds = b"".join(synth.key_seed(7_000_000 + i) for i in range(n))
    pks = _lib.public_keys(seeds, n)
    secrets = b"".join(seeds[32 * i:32 * i + 32] + pks[32 * i:32 * i + 32] for i in range(n))
    dg = synth.shared_digest(5) if shared else b"".join(synth.msg_digest(i) for i in range(n))
    return dg, pks, _lib.sign_batch(secrets, dg, n, shared=shared)


def measure(n, mode, kernel=_lib.KERNEL_AUTO, shared=True, nctx=4, calls=25, reps=8):
    dg, pk, sg = make(n, shared)
    d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
    ctxs = [_lib.DeviceContext(n) for _ in range(nctx)]
    for c in ctxs:
        c.set_rlc_min(1)
        if kernel != _lib.KERNEL_AUTO:
            c.set_kernel(kernel)
    bms = [torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev) for _ in range(nctx)]
    nrs = torch.zeros(nctx, dtype=torch.int32, device=dev)
    sts = [torch.cuda.ExternalStream(c.stream, device=dev) for c in ctxs]
    torch.cuda.synchronize(dev)
    stride = 0 if shared else 32

    def call(j):
        ctxs[j].verify(d_dg.data_ptr(), stride, d_pk.data_ptr(), d_sg.data_ptr(), n, mode, bms[j].data_ptr(), nrs[j:].data_ptr(), 0)
    lat = []
    for i in range(calls + 4):
        sts[0].synchronize(); t0 = time.perf_counter(); call(0); sts[0].synchronize()
        if i >= 4:
            lat.append(time.perf_counter() - t0)
    assert bool((bms[0].view(torch.int64)[: n // 64] == -1).all()) and int(nrs[0].item()) == 0
    if mode == _lib.BATCH_RLC:
        assert ctxs[0].rlc_last()[0] == 1
    # several contexts kept busy: the caller says so (throughput forms); HS_SWEEP_NO_HINT=1 measures without the hint
    if os.environ.get("HS_SWEEP_NO_HINT") != "1":
        for c in ctxs:
            c.set_concurrent(True)
    for j in range(nctx):
        call(j)
    for s in sts:
        s.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        for j in range(nctx):
            call(j)
    for s in sts:
        s.synchronize()
    wall = time.perf_counter() - t0
    for c in ctxs:
        c.close()
    lat.sort()
    return lat[len(lat) // 2] * 1e3, reps * nctx * n / wall / 1e6


print("# Round-2 sweeps (one MI355X, device-resident inputs, `tools/r2_sweeps.py`)\n")
print("## 1. Cold QC path: per-signature kernels vs batch RLC / MSM (HS_ED25519_BATCH_RLC)\n")
print("| signatures | per-signature p50 ms | RLC p50 ms | per-signature M sig/s (4 in flight) | RLC M sig/s (4 in flight) |")
print("|---|---|---|---|---|")
SIZES = [int(x) for x in os.environ["HS_SWEEP_SIZES"].split(",")] if os.environ.get("HS_SWEEP_SIZES") else \
    [2048, 4096, 6667, 8192, 12288, 16384, 24576, 32768, 49152, 66667, 131072, 262144]
for n in ([] if os.environ.get('HS_SWEEP_ONLY') == '3' else SIZES):
    a = measure(n, _lib.BATCH)
    b = measure(n, _lib.BATCH_RLC)
    print("| %d | %.3f | %.3f | %.1f | %.1f |" % (n, a[0], b[0], a[1], b[1]), flush=True)
print("\nlibrary threshold (hs_ed25519_rlc_min_batch): %d\n" % _lib.lib().hs_ed25519_rlc_min_batch())
SKIP12 = os.environ.get("HS_SWEEP_ONLY") == "3"
if os.environ.get("HS_SWEEP_SIZES"):
    sys.exit(0)
print("## 2. Per-signature cold path (strict, one digest per signature): fused four-lane kernel vs one lane per signature\n")
print("| signatures | k_verify_fused p50 ms | pairs form p50 ms | k_prepare + k_verify p50 ms | fused M sig/s (4 in flight) | pairs M sig/s | lane M sig/s |")
print("|---|---|---|---|---|---|---|")
for n in ([] if os.environ.get('HS_SWEEP_ONLY') == '3' else (8192, 16384, 32768, 49152, 66667, 131072, 262144)):
    a = measure(n, _lib.STRICT, _lib.KERNEL_FUSED, shared=False)
    c = measure(n, _lib.STRICT, _lib.KERNEL_FUSED_PAIRS, shared=False)
    b = measure(n, _lib.STRICT, _lib.KERNEL_LANE, shared=False)
    print("| %d | %.3f | %.3f | %.3f | %.1f | %.1f | %.1f |" % (n, a[0], c[0], b[0], a[1], c[1], b[1]), flush=True)

if os.environ.get("HS_SWEEP_SKIP3"):
    sys.exit(0)
print("\n## 3. Key-cache path (batch semantics, shared digest, one key per signature): two-lane + R-wavefront kernel vs chains + batched finish\n")
print("| signatures | lane2r p50 ms | chain2 + finish p50 ms | lane2r M sig/s (4 in flight) | chain2 + finish M sig/s (4 in flight) |")
print("|---|---|---|---|---|")


def measure_cached(n, kernel, nctx=4, calls=20, reps=8):
    dg, pk, sg = make(n, True)
    ks = _lib.KeySet([pk[32 * i:32 * i + 32] for i in range(n)])
    d_dg, d_sg = t(dg), t(sg)
    d_ki = torch.arange(n, dtype=torch.int32, device=dev)
    ctxs = [_lib.DeviceContext(n) for _ in range(nctx)]
    for c in ctxs:
        c.set_kernel(kernel)
    bms = [torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev) for _ in range(nctx)]
    nrs = torch.zeros(nctx, dtype=torch.int32, device=dev)
    sts = [torch.cuda.ExternalStream(c.stream, device=dev) for c in ctxs]
    torch.cuda.synchronize(dev)

    def call(j):
        ctxs[j].verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bms[j].data_ptr(), nrs[j:].data_ptr(), 0)
    lat = []
    for i in range(calls + 4):
        sts[0].synchronize(); t0 = time.perf_counter(); call(0); sts[0].synchronize()
        if i >= 4:
            lat.append(time.perf_counter() - t0)
    assert bool((bms[0].view(torch.int64)[: n // 64] == -1).all()) and int(nrs[0].item()) == 0
    for j in range(nctx):
        call(j)
    for s_ in sts:
        s_.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        for j in range(nctx):
            call(j)
    for s_ in sts:
        s_.synchronize()
    wall = time.perf_counter() - t0
    for c in ctxs:
        c.close()
    ks.close()
    lat.sort()
    return lat[len(lat) // 2] * 1e3, reps * nctx * n / wall / 1e6


for n in (1024, 6667, 16384, 66667, 131072, 262144):
    a = measure_cached(n, _lib.KERNEL_FUSED)
    b = measure_cached(n, _lib.KERNEL_CACHED_C)
    print("| %d | %.3f | %.3f | %.1f | %.1f |" % (n, a[0], b[0], a[1], b[1]), flush=True)
