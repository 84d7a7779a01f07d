"""Key cache: the three forms by size on one box - the latency kernel (two lanes per signature + a third wavefront decompressing
the R's: k_verify_cached_lane2r), the one-lane kernel (k_verify_cached_lane), and the two-launch "c" form (two-lane chains,
then four to sixteen signatures per finish lane compressed with one inversion: k_cached_chain2 + k_cached_finish) - p50 and
kernel time of a LONE device-resident call and signatures/s with four contexts kept busy (marked concurrent).  The batch-list
comparison of profiles/r4_j_cached_c_ab.md (the c form as the list default: 185 against 269 M verifies/s) was taken at commit
9323e7c, where the lists still defaulted to it.

    python tools/cached_c_ab.py [sweep]        (synthetic sets signed on the GPU; HS_SWEEP_SIZES=a,b,c)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
SIZES = [int(x) for x in os.environ.get("HS_SWEEP_SIZES", "1024,6667,16384,24576,32768,66667,131072,262144").split(",")]


def sweep():
    import torch
    import synth
    from hotstuff_amd import _lib
    _lib.init(0, 8192)
    dev = torch.device("cuda", 0)
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    nmax = max(SIZES)
    seeds = b"".join(synth.key_seed(7_000_000 + i) for i in range(nmax))
    pks = _lib.public_keys(seeds, nmax)
    secrets = b"".join(seeds[32 * i:32 * i + 32] + pks[32 * i:32 * i + 32] for i in range(nmax))
    dg = synth.shared_digest(5)
    sg = _lib.sign_batch(secrets, dg, nmax, shared=True)
    ks = _lib.KeySet([pks[32 * i:32 * i + 32] for i in range(nmax)])
    d_dg, d_sg = t(dg), t(sg)
    d_ki = torch.arange(nmax, dtype=torch.int32, device=dev)
    # (profiles/r4_j_cached_c_ab.md and r4_t_* carry a third column, "lane": the one-lane kernel k_verify_cached_lane, retired in round 5)
    forms = (("lane2r", _lib.KERNEL_FUSED), ("c", _lib.KERNEL_CACHED_C))
    print("| signatures | " + " | ".join("%s p50 / kernel ms" % f for f, _ in forms) + " | " + " | ".join("%s M sig/s, 4 in flight" % f for f, _ in forms) + " |")
    print("|---|" + "---|" * (2 * len(forms)))
    for n in SIZES:
        row_l, row_t = [], []
        for name, kernel in forms:
            nctx = 4
            ctxs = [_lib.DeviceContext(n) for _ in range(nctx)]
            for c in ctxs:
                c.set_kernel(kernel)
            ctxs[0].set_profiling(64)
            bms = [torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev) for _ in range(nctx)]
            nrs = torch.zeros(nctx, dtype=torch.int32, device=dev)
            sts = [torch.cuda.ExternalStream(c.stream, device=dev) for c in ctxs]
            torch.cuda.synchronize(dev)

            def call(j):
                ctxs[j].verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bms[j].data_ptr(), nrs[j:].data_ptr(), 0)
            lat = []
            for i in range(44):
                sts[0].synchronize(); t0 = time.perf_counter(); call(0); sts[0].synchronize()
                if i >= 4:
                    lat.append(time.perf_counter() - t0)
            assert bool((bms[0].view(torch.int64)[: n // 64] == -1).all()) and int(nrs[0].item()) == 0, (name, n)
            a, b, k = ctxs[0].kernel_times()
            ctxs[0].set_profiling(0)
            for c in ctxs:                  # several kept busy: the caller says so (the finish then takes 8 / 16 signatures per lane instead of 4)
                c.set_concurrent(True)
            for j in range(nctx):
                call(j)
            for s_ in sts:
                s_.synchronize()
            reps = 8
            t0 = time.perf_counter()
            for i in range(reps):
                for j in range(nctx):
                    call(j)
            for s_ in sts:
                s_.synchronize()
            wall = time.perf_counter() - t0
            for c in ctxs:
                c.close()
            lat.sort()
            row_l.append("%.4f / %.4f" % (lat[len(lat) // 2] * 1e3, (a + b) / max(1, k)))
            row_t.append("%.1f" % (reps * nctx * n / wall / 1e6))
        print("| %d | %s | %s |" % (n, " | ".join(row_l), " | ".join(row_t)), flush=True)
    ks.close()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        return sweep()
    print("# key cache: latency kernel / one-lane kernel / chains + batched finish by compression, same box\n")
    o = subprocess.run([sys.executable, os.path.abspath(__file__), "sweep"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    print(o.stdout if o.returncode == 0 else "sweep FAILED: " + (o.stderr or o.stdout)[-800:], flush=True)


if __name__ == "__main__":
    main()
