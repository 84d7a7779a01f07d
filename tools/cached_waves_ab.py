"""Key-cache kernel k_verify_cached_lane2r compiled for 3 wavefronts per SIMD (165 VGPRs, no scratch: the product) against 4
(128 VGPRs, 216 B of scratch per lane; build/variant/libhs_cached4.so): does a fourth resident wavefront absorb the 18-workgroup
second round of the 66,667-signature QC (1,042 workgroups for 1,024 slots)?  VERDICT r3 "Next round" #6b.

    python tools/cached_waves_ab.py        (driver; alternates the two libraries in subprocesses)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
SIZES = (6667, 65536, 66667, 131072)


def worker():
    import torch
    import synth
    from hotstuff_amd import _lib
    dev = torch.device("cuda:0")
    _lib.init(0, 8192)
    out = {}
    nmax = max(SIZES)
    dg, pk, sg = synth.make_batch(nmax, shared=True, cache_dir="/tmp/hs_synth")
    ks = _lib.KeySet([pk[32 * i:32 * i + 32] for i in range(nmax)])
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_dg, d_sg = t(dg), t(sg)
    d_ki = torch.arange(nmax, dtype=torch.int32, device=dev)
    for n in SIZES:
        bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
        nr = torch.zeros(1, dtype=torch.int32, device=dev)
        ctx = _lib.DeviceContext(n)
        ctx.set_profiling(128)
        st = torch.cuda.ExternalStream(ctx.stream, device=dev)
        torch.cuda.synchronize(dev)
        ts = []
        with torch.cuda.stream(st):
            for i in range(110):
                nr.zero_(); st.synchronize()
                t0 = time.perf_counter()
                ctx.verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), nr.data_ptr(), ctx.stream)
                bad = int(nr.item())
                dt = time.perf_counter() - t0
                assert bad == 0
                if i >= 10:
                    ts.append(dt)
        ts.sort()
        a, b, k = ctx.kernel_times()
        out[str(n)] = {"p50_ms": round(ts[50] * 1e3, 4), "kernel_ms": round(b / max(1, k), 4)}
        ctx.close()
    ks.close()
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        return worker()
    libs = {"3 wavefronts / SIMD (165 VGPRs)": os.path.join(ROOT, "hotstuff_amd", "libhs_ed25519.so"),
            "4 wavefronts / SIMD (128 VGPRs, 216 B scratch)": os.path.join(ROOT, "build", "variant", "libhs_cached4.so")}
    res = {k: [] for k in libs}
    for rnd in range(2):
        for name, path in libs.items():
            if not os.path.exists(path):
                print("missing", path); continue
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=dict(os.environ, HS_ED25519_LIB=path), capture_output=True, text=True, timeout=400)
            line = [l for l in o.stdout.splitlines() if l.startswith("{")]
            if line:
                res[name].append(json.loads(line[-1]))
            else:
                print("worker failed:", o.stderr[-400:])
    print("| signatures | " + " | ".join(libs) + " |   (p50 / kernel ms per run)")
    print("|---|---|---|")
    for n in SIZES:
        print("| %d | %s |" % (n, " | ".join(" ; ".join("%.4f / %.4f" % (r[str(n)]["p50_ms"], r[str(n)]["kernel_ms"]) for r in res[k]) for k in libs)))


if __name__ == "__main__":
    main()
