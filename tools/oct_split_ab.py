"""A/B of the lone-call cold form for n <= 8,192: eight lanes per signature on ONE wavefront (k_verify_fused8, HS_ED25519_OCT_SPLIT=0)
against the same split over a main and a helper wavefront (k_verify_fused8h, the default).  Device-resident p50 of a lone call and
the kernel's own dispatch-timestamp duration; alternating subprocesses on the same box.

    python tools/oct_split_ab.py            (driver)        python tools/oct_split_ab.py worker   (one configuration, env decides)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
SIZES = (64, 1024, 4096, 6667, 8192)


def worker():
    import torch
    import synth
    from hotstuff_amd import _lib
    dev = torch.device("cuda:0")
    _lib.init(0, 8192)
    out = {"kernel": _lib.lib().hs_ed25519_dominant_kernel_for(6667).decode(), "rows": {}}
    for n in SIZES:
        dg, pk, sg = synth.make_batch(n, shared=True, cache_dir="/tmp/hs_synth")
        t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
        bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
        nr = torch.zeros(1, dtype=torch.int32, device=dev)
        ctx = _lib.DeviceContext(n)
        ctx.set_profiling(256)
        st = torch.cuda.ExternalStream(ctx.stream, device=dev)
        torch.cuda.synchronize(dev)
        ts = []
        with torch.cuda.stream(st):
            for i in range(220):
                nr.zero_()
                st.synchronize()
                t0 = time.perf_counter()
                ctx.verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), n, _lib.STRICT, bm.data_ptr(), nr.data_ptr(), ctx.stream)
                bad = int(nr.item())
                dt = time.perf_counter() - t0
                assert bad == 0
                if i >= 20:
                    ts.append(dt)
        ts.sort()
        a, b, k = ctx.kernel_times()
        out["rows"][str(n)] = {"p50_ms": round(ts[len(ts) // 2] * 1e3, 4), "p10_ms": round(ts[len(ts) // 10] * 1e3, 4),
                               "kernel_ms": round(b / max(1, k), 4)}
        ctx.close()
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        return worker()
    res = {"0": [], "1": []}
    for rnd in range(3):
        for split in ("0", "1"):
            env = dict(os.environ, HS_ED25519_OCT_SPLIT=split)
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=env, capture_output=True, text=True, timeout=300)
            line = [l for l in o.stdout.splitlines() if l.startswith("{")]
            if not line:
                print("worker failed:", o.stderr[-500:]); continue
            res[split].append(json.loads(line[-1]))
    print("| signatures | one wavefront (k_verify_fused8): p50 / kernel ms | split (k_verify_fused8h): p50 / kernel ms |")
    print("|---|---|---|")
    for n in SIZES:
        cell = lambda rs: " ; ".join("%.4f / %.4f" % (r["rows"][str(n)]["p50_ms"], r["rows"][str(n)]["kernel_ms"]) for r in rs)
        print("| %d | %s | %s |" % (n, cell(res["0"]), cell(res["1"])))
    print("kernels:", sorted({r["kernel"] for r in res["0"]}), sorted({r["kernel"] for r in res["1"]}))


if __name__ == "__main__":
    main()
