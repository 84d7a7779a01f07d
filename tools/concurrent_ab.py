"""Same-box A/B of hs_ed25519_ctx_set_concurrent: cold QC-shaped calls (shared digest, batch semantics), device-resident,
four contexts.  Per size: p50 of one call at a time and signatures/s with four calls in flight, per-signature and RLC, with the
hint off (latency forms: eight lanes per signature, side-stream sliver) and on (throughput forms).

    python tools/concurrent_ab.py [sizes]           -> markdown rows
"""
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


def measure(n, mode, hint, nctx=4, calls=25, reps=8):
    seeds = b"".join(synth.key_seed(7_000_000 + i) for i in range(n))
    pk = _lib.public_keys(seeds, n)
    secrets = b"".join(seeds[32 * i:32 * i + 32] + pk[32 * i:32 * i + 32] for i in range(n))
    dg = synth.shared_digest(3)
    sg = _lib.sign_batch(secrets, dg, n, shared=True)
    d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
    ctxs = [_lib.DeviceContext(n) for _ in range(nctx)]
    for c in ctxs:
        c.set_rlc_min(1)
        c.set_concurrent(hint)
    bms = [torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev) for _ in range(nctx)]
    nrs = torch.zeros(nctx, dtype=torch.int32, device=dev)
    sts = [torch.cuda.ExternalStream(c.stream, device=dev) for c in ctxs]
    torch.cuda.synchronize(dev)

    def call(j):
        ctxs[j].verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), n, mode, bms[j].data_ptr(), nrs[j:].data_ptr(), 0)
    lat = []
    for i in range(calls + 4):
        sts[0].synchronize(); t0 = time.perf_counter(); call(0); sts[0].synchronize()
        if i >= 4:
            lat.append(time.perf_counter() - t0)
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
    for c in ctxs:
        c.close()
    lat.sort()
    return lat[len(lat) // 2] * 1e3, reps * nctx * n / wall / 1e6


sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4096,6667,66667,131072").split(",")]
print("| signatures | hint | per-signature p50 ms | RLC p50 ms | per-signature M sig/s, 4 in flight | RLC M sig/s, 4 in flight |")
print("|---|---|---|---|---|---|")
for rnd in range(2):            # two alternating rounds
    for n in sizes:
        for hint in (False, True):
            p, pt = measure(n, _lib.BATCH, hint)
            r, rt = measure(n, _lib.BATCH_RLC, hint)
            print("| %d | %s | %.3f | %.3f | %.1f | %.1f |" % (n, "on" if hint else "off", p, r, pt, rt), flush=True)
_lib.shutdown()
