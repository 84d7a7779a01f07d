"""Which class of box is this?  The QC of a 10,000-validator committee (6,667 signatures), key cache and cold, 200 lone calls
each: host-visible p50 and the kernel's own time from its dispatch timestamps - the two numbers every other record of
tools/gpu_r4.sh's env_record is read against.  One JSON line.   python tools/qc_marker.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import torch
    import synth
    from hotstuff_amd import _lib
    n, K = 6667, 10000
    dev = torch.device("cuda:0")
    _lib.init(0, 8192)
    dg, pk, sg = synth.make_batch(K, shared=True, cache_dir="/tmp/hs_synth")
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    idx = [(i * K) // n for i in range(n)]
    d_dg, d_pk = t(dg), t(b"".join(pk[32 * j:32 * j + 32] for j in idx))
    d_sg = t(b"".join(sg[64 * j:64 * j + 64] for j in idx))
    d_ki = torch.tensor(idx, dtype=torch.int32, device=dev)
    ks = _lib.KeySet([pk[32 * i:32 * i + 32] for i in range(K)])
    out = {}
    for form in ("cached", "cold"):
        bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
        nr = torch.zeros(1, dtype=torch.int32, device=dev)
        ctx = _lib.DeviceContext(n)
        ctx.set_profiling(256)
        st = torch.cuda.ExternalStream(ctx.stream, device=dev)
        torch.cuda.synchronize(dev)
        ts = []
        with torch.cuda.stream(st):
            for i in range(220):
                nr.zero_(); st.synchronize()
                t0 = time.perf_counter()
                if form == "cached":
                    ctx.verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), nr.data_ptr(), ctx.stream)
                else:
                    ctx.verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), nr.data_ptr(), ctx.stream)
                bad = int(nr.item())
                dt = time.perf_counter() - t0
                assert bad == 0, bad
                if i >= 20:
                    ts.append(dt)
        ts.sort()
        a, b, k = ctx.kernel_times()
        out[form] = {"p50_ms": round(ts[100] * 1e3, 4), "p10_ms": round(ts[20] * 1e3, 4), "kernel_ms": round((a + b) / max(1, k), 4)}
        ctx.close()
    ks.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
