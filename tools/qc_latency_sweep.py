"""QC::verify-equivalent p50 latency versus committee size N (n = floor(2N/3) + 1 signatures over one
digest, batch semantics): cold path and committee key cache, through the host-pointer C-ABI
(PCIe both ways included) and device-resident.  Writes a markdown table to stdout.

    python tools/qc_latency_sweep.py [--max 100000]
"""
import argparse
import ctypes
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max", type=int, default=100000)
    ap.add_argument("--calls", type=int, default=200)
    args = ap.parse_args()
    import torch
    import synth
    from hotstuff_amd import _lib
    dev = torch.device("cuda:0")
    _lib.init(0, 8192)
    lib = _lib.lib()
    med = lambda v: sorted(v)[len(v) // 2] * 1e3
    print("| N validators | signatures | cold host API | cold device | cached host API | cached device | key set |")
    print("|---|---|---|---|---|---|---|")
    for N in [4, 10, 50, 100, 1000, 10000, 30000, 100000]:
        if N > args.max:
            break
        n = 2 * N // 3 + 1
        dg, pk, sg = synth.make_batch(n, shared=True, cache_dir="/tmp/hs_synth")
        b_dg, b_pk, b_sg = (ctypes.create_string_buffer(x, len(x)) for x in (dg, pk, sg))
        idx = struct.pack("<%dI" % n, *range(n))
        b_ix = ctypes.create_string_buffer(idx, len(idx))
        keys = [pk[32 * i:32 * i + 32] for i in range(n)] + [synth.key_seed(400_000_000 + i) for i in range(N - n)]
        ks = _lib.KeySet(keys)
        calls = args.calls if n < 20000 else 60

        def host(fn):
            ts = []
            for i in range(calls + 10):
                t0 = time.perf_counter()
                rc = fn()
                dt = time.perf_counter() - t0
                assert rc == 1
                if i >= 10:
                    ts.append(dt)
            return med(ts)

        h_cold = host(lambda: lib.hs_ed25519_verify_shared(ctypes.addressof(b_dg), ctypes.addressof(b_pk),
                                                           ctypes.addressof(b_sg), n, _lib.BATCH, None))
        h_cach = host(lambda: lib.hs_ed25519_verify_shared_cached(ks.handle, ctypes.addressof(b_dg), ctypes.addressof(b_ix),
                                                                  ctypes.addressof(b_sg), n, _lib.BATCH, None))
        t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
        d_ki = torch.arange(n, dtype=torch.int32, device=dev)
        bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
        nr = torch.zeros(1, dtype=torch.int32, device=dev)
        ctx = _lib.DeviceContext(n)
        st = torch.cuda.ExternalStream(ctx.stream, device=dev)
        torch.cuda.synchronize(dev)

        def devrun(cached):
            ts = []
            with torch.cuda.stream(st):
                for i in range(calls + 10):
                    nr.zero_()
                    st.synchronize()
                    t0 = time.perf_counter()
                    if cached:
                        ctx.verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH,
                                          bm.data_ptr(), nr.data_ptr(), ctx.stream)
                    else:
                        ctx.verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(),
                                   nr.data_ptr(), ctx.stream)
                    bad = int(nr.item())
                    dt = time.perf_counter() - t0
                    assert bad == 0
                    if i >= 10:
                        ts.append(dt)
            return med(ts)

        d_cold, d_cach = devrun(False), devrun(True)
        print("| %d | %d | %.3f ms | %.3f ms | %.3f ms | %.3f ms | %.2f GB |"
              % (N, n, h_cold, d_cold, h_cach, d_cach, ks.hbm_bytes() / 1e9), flush=True)
        ctx.close(); ks.close()


if __name__ == "__main__":
    main()
