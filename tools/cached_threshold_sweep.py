"""Which key-cache kernel wins at which batch size: device-resident p50 of one call, forced kernels.
  python tools/cached_threshold_sweep.py > gpurun_out/cached_threshold_sweep.md     (on a GPU box)
Keys: min(n, 20000) registered, signature i signed by key i mod that (tables up to 13 GB)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402
from hotstuff_amd import _lib  # noqa: E402

CACHE = os.environ.get("HS_SYNTH_CACHE", "/tmp/hs_synth")
KINDS = [("4 lanes + R wavefront (5w)", _lib.KERNEL_FUSED), ("2 lanes + R wavefront", _lib.KERNEL_CACHED_LANE2_R),
         ("2 lanes, inversion", _lib.KERNEL_QUAD), ("1 lane, inversion", _lib.KERNEL_LANE)]


def main():
    _lib.init(0, 8192)
    dev = torch.device("cuda:0")
    nk = 20000
    dg, pk, sg = synth.make_batch(nk, shared=True, cache_dir=CACHE)
    ks = _lib.KeySet([pk[32 * i:32 * i + 32] for i in range(nk)])
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_dg = t(dg)
    print("| signatures | " + " | ".join(k for k, _ in KINDS) + " |")
    print("|---|" + "---|" * len(KINDS))
    sizes = [int(x) for x in sys.argv[1:]] or [4096, 8192, 12288, 16384, 20001, 32768, 66667, 131072, 200000, 262144]
    for n in sizes:
        reps = (n + nk - 1) // nk
        d_sg = t((sg * reps)[: 64 * n])
        d_ki = (torch.arange(n, dtype=torch.int32, device=dev) % nk).contiguous()
        bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
        nr = torch.zeros(1, dtype=torch.int32, device=dev)
        ctx = _lib.DeviceContext(n)
        st = torch.cuda.ExternalStream(ctx.stream, device=dev)
        torch.cuda.synchronize(dev)
        row = []
        for _, sel in KINDS:
            old = _lib.set_kernel(sel)
            ts = []
            with torch.cuda.stream(st):
                for i in range(25):
                    nr.zero_()
                    st.synchronize()
                    t0 = time.perf_counter()
                    ctx.verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(),
                                      nr.data_ptr(), ctx.stream)
                    bad = int(nr.item())
                    ts.append(time.perf_counter() - t0)
                    assert bad == 0
            _lib.set_kernel(old)
            ts = sorted(ts[5:])
            row.append("%.3f ms" % (ts[len(ts) // 2] * 1e3))
        ctx.close()
        print("| %d | %s |" % (n, " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
