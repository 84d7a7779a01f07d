"""Key-cache QC of 6,667 signatures against key sets of 667 ... 20,000 registered keys (0.35 ... 10.6 GB of tables): does the
kernel's time follow the FOOTPRINT its 32 per-key lookups are spread over?  Round 4's two "classes of box" have the same
lone-wavefront clock and the same dependent-gather latency up to 2 GB; the key table of a 10,000-validator committee is
5.3 GB.  If the slow class is slow only on the large tables, it is the memory system's reach (TLB / page fragments), not the
kernel.   python tools/cached_footprint_probe.py [n_signatures]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import torch
    import synth
    from hotstuff_amd import _lib
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6667
    dev = torch.device("cuda:0")
    _lib.init(0, 8192)
    kmax = 20000
    dg, pk, sg = synth.make_batch(kmax, shared=True, cache_dir="/tmp/hs_synth")
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_dg = t(dg)
    print("| registered keys | table GB | p50 ms | kernel ms | p10 | p99 |")
    print("|---|---|---|---|---|---|")
    for K in (667, 2000, 6667, 10000, 20000, 667):
        ks = _lib.KeySet([pk[32 * i:32 * i + 32] for i in range(K)])
        idx = [i % K for i in range(n)]
        # spread the n signatures over the whole key set when it is larger than n (a QC of 2f+1 of N validators)
        if K > n:
            idx = [(i * K) // n for i in range(n)]
        d_sg = t(b"".join(sg[64 * j:64 * j + 64] for j in idx))
        d_ki = torch.tensor(idx, dtype=torch.int32, device=dev)
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
                ctx.verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), nr.data_ptr(), ctx.stream)
                bad = int(nr.item())
                dt = time.perf_counter() - t0
                assert bad == 0, bad
                if i >= 20:
                    ts.append(dt)
        ts.sort()
        a, b, k = ctx.kernel_times()
        print("| %d | %.2f | %.4f | %.4f | %.4f | %.4f |" % (K, K * 528384 / 1e9, ts[100] * 1e3, b / max(1, k), ts[20] * 1e3, ts[197] * 1e3), flush=True)
        ctx.close()
        ks.close()
        del d_sg, d_ki


if __name__ == "__main__":
    main()
