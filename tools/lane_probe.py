"""Throughput of the one-lane cold pair (k_prepare + k_verify) on large strict batches, four calls in flight
(the shape of tools/r2_sweeps.py section 2).   python tools/lane_probe.py [n ...]"""
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
for n in [int(x) for x in sys.argv[1:]] or [66667, 262144]:
    seeds = b"".join(synth.key_seed(7_000_000 + i) for i in range(n))
    pk = _lib.public_keys(seeds, n)
    secrets = b"".join(seeds[32 * i:32 * i + 32] + pk[32 * i:32 * i + 32] for i in range(n))
    dg = b"".join(synth.msg_digest(i) for i in range(n))
    sg = _lib.sign_batch(secrets, dg, n, shared=False)
    d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
    nctx, reps = 4, 8
    ctxs = [_lib.DeviceContext(n) for _ in range(nctx)]
    for c in ctxs:
        c.set_kernel(_lib.KERNEL_LANE)
    bms = [torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev) for _ in range(nctx)]
    nrs = torch.zeros(nctx, dtype=torch.int32, device=dev)
    sts = [torch.cuda.ExternalStream(c.stream, device=dev) for c in ctxs]
    torch.cuda.synchronize(dev)
    call = lambda j: ctxs[j].verify(d_dg.data_ptr(), 32, d_pk.data_ptr(), d_sg.data_ptr(), n, _lib.STRICT, bms[j].data_ptr(), nrs[j:].data_ptr(), 0)
    lat = []
    for i in range(12):
        sts[0].synchronize(); t0 = time.perf_counter(); call(0); sts[0].synchronize(); lat.append(time.perf_counter() - t0)
    assert bool((bms[0].view(torch.int64)[: n // 64] == -1).all()) and int(nrs[0].item()) == 0
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
    print("lane kernels n=%d: p50 %.3f ms, %.1f M sig/s with 4 calls in flight" % (n, sorted(lat[4:])[4] * 1e3, reps * nctx * n / wall / 1e6), flush=True)
    for c in ctxs:
        c.close()
