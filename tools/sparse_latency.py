"""QC::verify latency when certificates arrive as a node's do - one per round, with idle time in between - instead of back to back.

bench.py's `qc_verify_p50_ms_*` is the p50 over 200 calls issued one after the other: the GPU's clocks are up, the kernel's code
and tables are in the caches.  /root/reference/consensus/src/core.rs:379 verifies ONE certificate per round; rounds are
milliseconds (LAN, `fab local`) to hundreds of milliseconds (WAN) apart.  This measures what such a caller sees: for every gap
in --gaps (milliseconds of sleep between calls; 0 = back to back) the p50 / p90 / max of
    host     hs_ed25519_verify_shared[_cached]: host buffers in, verdict out (what the shim calls)
    kernel   the dispatch-timestamp duration of the verification kernel(s) of the same calls issued device-resident
for the 6,667-signature QC of a 10,000-validator committee, cold and through the key cache.  With --keep-warm MS a second thread
verifies a 64-signature batch every MS milliseconds - what a node's own trickle of small verifications (votes, timeouts) does.

    python tools/sparse_latency.py [--gaps 0,1,10,100,1000] [--calls 40] [--keep-warm 0] > profiles/r5_sparse_latency.md
"""
import argparse
import ctypes
import os
import struct
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaps", default="0,1,10,100,1000")
    ap.add_argument("--calls", type=int, default=40)
    ap.add_argument("--validators", type=int, default=10000)
    ap.add_argument("--keep-warm", type=float, default=0.0, help="another thread verifies 64 signatures every this many ms (0 = nobody)")
    args = ap.parse_args()
    import torch
    import synth
    from hotstuff_amd import _lib
    dev = torch.device("cuda:0")
    _lib.init(0, 8192)
    lib = _lib.lib()
    N = args.validators
    n = 2 * N // 3 + 1
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir="/tmp/hs_synth")
    b_dg, b_pk, b_sg = (ctypes.create_string_buffer(x, len(x)) for x in (dg, pk, sg))
    idx = struct.pack("<%dI" % n, *range(n))
    b_ix = ctypes.create_string_buffer(idx, len(idx))
    ks = _lib.KeySet([pk[32 * i:32 * i + 32] for i in range(n)])
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
    d_ki = torch.arange(n, dtype=torch.int32, device=dev)
    bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
    ctx = _lib.DeviceContext(n)
    st = torch.cuda.ExternalStream(ctx.stream, device=dev)
    torch.cuda.synchronize(dev)

    stop = threading.Event()
    warm_calls = [0]

    def keep_warm():
        m = 64
        w_dg, w_pk, w_sg = (ctypes.create_string_buffer(x, len(x)) for x in (dg, pk[:32 * m], sg[:64 * m]))
        while not stop.is_set():
            assert lib.hs_ed25519_verify_shared(ctypes.addressof(w_dg), ctypes.addressof(w_pk), ctypes.addressof(w_sg), m, _lib.BATCH, None) == 1
            warm_calls[0] += 1
            time.sleep(args.keep_warm / 1e3)

    th = None
    if args.keep_warm > 0:
        th = threading.Thread(target=keep_warm, daemon=True)
        th.start()

    def host(cached):
        if cached:
            return lib.hs_ed25519_verify_shared_cached(ks.handle, ctypes.addressof(b_dg), ctypes.addressof(b_ix), ctypes.addressof(b_sg), n, _lib.BATCH, None)
        return lib.hs_ed25519_verify_shared(ctypes.addressof(b_dg), ctypes.addressof(b_pk), ctypes.addressof(b_sg), n, _lib.BATCH, None)

    def device(cached):
        if cached:
            ctx.verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), 0, ctx.stream)
        else:
            ctx.verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), 0, ctx.stream)
        st.synchronize()

    def stats(v):
        v = sorted(v)
        return v[len(v) // 2], v[(9 * len(v)) // 10], v[-1]

    print("QC of N = %d validators (%d signatures), %d calls per cell after 3 untimed ones; ms; keep-warm thread: %s\n"
          % (N, n, args.calls, ("64 signatures every %g ms" % args.keep_warm) if th else "none"))
    print("| gap between calls | cold host p50 / p90 / max | cold kernel p50 / max | key cache host p50 / p90 / max | key cache kernel p50 / max |")
    print("|---|---|---|---|---|")
    for gap in [float(g) for g in args.gaps.split(",")]:
        calls = args.calls if gap < 500 else max(8, args.calls // 4)
        row = []
        for cached in (False, True):
            hs, ks_ms = [], []
            for i in range(calls + 3):
                if gap:
                    time.sleep(gap / 1e3)
                t0 = time.perf_counter()
                rc = host(cached)
                dt = (time.perf_counter() - t0) * 1e3
                assert rc == 1
                if i >= 3:
                    hs.append(dt)
            for i in range(calls + 3):
                if gap:
                    time.sleep(gap / 1e3)
                ctx.set_profiling(1)
                device(cached)
                a, b, k = ctx.kernel_times()
                if i >= 3 and k:
                    ks_ms.append(a + b)
            ctx.set_profiling(0)
            h, k = stats(hs), stats(ks_ms)
            row += ["%.3f / %.3f / %.3f" % h, "%.3f / %.3f" % (k[0], k[2])]
        print("| %s | %s |" % ("back to back" if not gap else "%g ms" % gap, " | ".join(row)), flush=True)
    stop.set()
    if th:
        th.join(5)
        print("\nkeep-warm calls made: %d" % warm_calls[0])
    ctx.close()
    ks.close()


if __name__ == "__main__":
    main()
