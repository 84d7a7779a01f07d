"""Where does a key-cache QC's time go, wavefront by wavefront?  Runs the trace build of k_verify_cached_lane2r
(build/variant/libhs_trace.so: -DHS_CACHED_TRACE, every wavefront stamps the constant-rate clock at its phase boundaries) on QCs
of 64 ... 6,667 signatures against a 10,000-key set and prints, per phase, the median and the maximum over the wavefronts.
Which phase stretches when the grid grows from one workgroup to 105 - the pure arithmetic (scalars, the R wavefront's square
root) or the table walk?  And between two boxes?        python tools/cached_trace.py"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
LIB = os.path.join(ROOT, "build", "variant", "libhs_trace.so")


def worker():
    import torch
    import synth
    from hotstuff_amd import _lib
    K = 10000
    dev = torch.device("cuda:0")
    _lib.init(0, 8192)
    setter = _lib.lib().hs_debug_set_cached_trace
    setter.argtypes = [ctypes.c_void_p]; setter.restype = ctypes.c_int
    dg, pk, sg = synth.make_batch(K, shared=True, cache_dir="/tmp/hs_synth")
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    ks = _lib.KeySet([pk[32 * i:32 * i + 32] for i in range(K)])
    d_dg = t(dg)
    wall_khz = 100000.0
    out = {}
    for n in (64, 1024, 6667, 64):
        idx = [(i * K) // n for i in range(n)]
        d_sg = t(b"".join(sg[64 * j:64 * j + 64] for j in idx))
        d_ki = torch.tensor(idx, dtype=torch.int32, device=dev)
        bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
        nr = torch.zeros(1, dtype=torch.int32, device=dev)
        blocks = (n + 63) // 64
        trace = torch.zeros(blocks * 3 * 8, dtype=torch.int64, device=dev)
        assert setter(trace.data_ptr()) == 0
        ctx = _lib.DeviceContext(n)
        ctx.set_profiling(64)
        st = torch.cuda.ExternalStream(ctx.stream, device=dev)
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(st):
            for i in range(30):
                ctx.verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), nr.data_ptr(), ctx.stream)
                st.synchronize()
        assert int(nr.item()) == 0
        a, b, k = ctx.kernel_times()
        tr = trace.cpu().view(blocks * 3, 8).tolist()
        us = lambda ticks: ticks / wall_khz * 1e3
        ph = {"scalars": [], "chain": [], "chain_wait": [], "finish": [], "sqrt": [], "sqrt_wait": []}
        t0s, t4s = [], []
        for w, r in enumerate(tr):
            t0s.append(r[0])
            if w % 3 == 2:
                ph["sqrt"].append(us(r[2] - r[0])); ph["sqrt_wait"].append(us(r[3] - r[2])); t4s.append(r[3])
            else:
                ph["scalars"].append(us(r[1] - r[0])); ph["chain"].append(us(r[2] - r[1])); ph["chain_wait"].append(us(r[3] - r[2]))
                ph["finish"].append(us(r[4] - r[3])); t4s.append(r[4])
        row = {"kernel_ms": round(b / max(1, k), 4), "span_us": round(us(max(t4s) - min(t0s)), 1), "start_skew_us": round(us(max(t0s) - min(t0s)), 1)}
        for name, v in ph.items():
            v.sort()
            row[name] = [round(v[len(v) // 2], 1), round(v[-1], 1)]
        out["%d%s" % (n, "_again" if str(n) in out else "")] = row
        setter(None)
        ctx.close()
    ks.close()
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        return worker()
    if not os.path.exists(LIB):
        print("missing", LIB); return
    o = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=dict(os.environ, HS_ED25519_LIB=LIB), capture_output=True, text=True, timeout=300)
    line = [l for l in o.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("worker failed:", o.stderr[-800:]); return
    res = json.loads(line[-1])
    print("| signatures | kernel ms | first start -> last end us | start skew | scalars (median / max us) | chain | chain waits at barrier | finish | R wavefront: square root | R waits |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for n, r in res.items():
        f = lambda k: "%.1f / %.1f" % tuple(r[k])
        print("| %s | %.4f | %.1f | %.1f | %s | %s | %s | %s | %s | %s |" % (n, r["kernel_ms"], r["span_us"], r["start_skew_us"], f("scalars"), f("chain"), f("chain_wait"),
                                                                            f("finish"), f("sqrt"), f("sqrt_wait")))


if __name__ == "__main__":
    main()
