"""Does the QC-verify p50 depend on the benchmark verifying the SAME certificate over and over?

bench.py's `qc_verify_p50_ms_*` re-verifies one 6,667-signature batch 200 times.  The table lines a verification reads are chosen
by its scalars - [v s]B from the 67 MB base-point table (16 lines per signature), and through the key cache 48 lines per
signature out of the committee's 5.3 GB of key tables - so a repeated batch finds yesterday's lines in the L2 / the 256 MB
last-level cache, and a node never verifies the same certificate twice.  This times the same call over a ROTATION of M distinct
certificates of one committee (same 10,000 keys, another round's digest, hence other signatures, other hashes, other table lines):
M = 1 is the bench's regime; M x 41 MB of key-table lines exceed the last-level cache from M = 8.  Device-resident calls, kernel
time from dispatch timestamps and the call's wall time, cold and key cache.

    python tools/fresh_inputs.py [--rotations 1,2,8,32] [--calls 200] > profiles/r5_fresh_inputs.md
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rotations", default="1,2,8,32")
    ap.add_argument("--calls", type=int, default=200)
    ap.add_argument("--validators", type=int, default=10000)
    ap.add_argument("--kernel", default="auto", help="auto | cached_c (the two-launch key-cache form)")
    args = ap.parse_args()
    import torch
    import synth
    from hotstuff_amd import _lib
    dev = torch.device("cuda:0")
    _lib.init(0, 8192)
    N = args.validators
    n = 2 * N // 3 + 1
    rots = [int(x) for x in args.rotations.split(",")]
    M = max(rots)
    # the committee: N keys derived on the GPU; certificate r = the first n validators' signatures over round r's digest
    seeds = b"".join(synth.key_seed(7_000_000 + i) for i in range(N))
    pks = _lib.public_keys(seeds, N)
    secrets = b"".join(seeds[32 * i:32 * i + 32] + pks[32 * i:32 * i + 32] for i in range(n))
    t0 = time.time()
    certs = []
    for r in range(M):
        dg = synth.shared_digest(1000 + r)
        certs.append((dg, _lib.sign_batch(secrets, dg, n, shared=True)))
    print("QC of N = %d validators (%d signatures); %d distinct certificates signed on the GPU in %.1f s; %d calls per cell; ms\n"
          % (N, n, M, time.time() - t0, args.calls))
    ks = _lib.KeySet([pks[32 * i:32 * i + 32] for i in range(N)])
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_pk = t(pks[:32 * n])
    d_ki = torch.arange(n, dtype=torch.int32, device=dev)
    d_certs = [(t(dg), t(sg)) for dg, sg in certs]
    bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
    nr = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx = _lib.DeviceContext(n)
    if args.kernel == "cached_c":
        ctx.set_kernel(_lib.KERNEL_CACHED_C)
    st = torch.cuda.ExternalStream(ctx.stream, device=dev)
    torch.cuda.synchronize(dev)

    def run(cached, m):
        walls, kerns = [], []
        for i in range(args.calls + 10):
            d_dg, d_sg = d_certs[i % m]
            ctx.set_profiling(1)
            t0 = time.perf_counter()
            if cached:
                ctx.verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), nr.data_ptr(), ctx.stream)
            else:
                ctx.verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), nr.data_ptr(), ctx.stream)
            st.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            a, b, k = ctx.kernel_times()
            if i >= 10:
                walls.append(dt)
                kerns.append(a + b)
        assert int(nr.item()) == 0, "a valid certificate was rejected"
        med = lambda v: sorted(v)[len(v) // 2]
        p99 = lambda v: sorted(v)[(99 * len(v)) // 100]
        return med(walls), med(kerns), p99(kerns)

    print("| distinct certificates in rotation | key-table lines touched per rotation | cold: call p50 / kernel p50 / kernel p99 | key cache: call p50 / kernel p50 / kernel p99 |")
    print("|---|---|---|---|")
    for m in rots:
        c = run(False, m)
        k = run(True, m)
        print("| %d%s | %.0f MB | %.3f / %.3f / %.3f | %.3f / %.3f / %.3f |" % (m, " (the bench's regime)" if m == 1 else "", m * n * 48 * 128 / 1e6, *c, *k), flush=True)
    ctx.close()
    ks.close()


if __name__ == "__main__":
    main()
