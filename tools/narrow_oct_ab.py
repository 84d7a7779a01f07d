"""The two eight-lane cold forms on a NARROW context (a stream confined to one 32-CU slice, what the host API gives calls of up to
1,024 signatures) against the whole chip: hs_ed25519_autotune measures the forms on the whole chip with a 6,667-signature QC;
does its pick transfer to a slice, where a 1,024-signature call is two wavefronts per SIMD in the split form?
    python tools/narrow_oct_ab.py > profiles/r5_y_narrow_oct_ab.md"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import synth  # noqa: E402
from hotstuff_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
_lib.init(0, 8192)
print("autotune (whole chip, 6,667 signatures):", _lib.config_dump()["devices"][0], "\n")
print("| signatures | context | k_verify_fused8 kernel / call ms | k_verify_fused8h kernel / call ms | k_verify_fused (four lanes) kernel / call ms | the library's own choice (selector 0) |")
print("|---|---|---|---|---|---|")
t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
for n in (256, 512, 667, 1024):
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=os.environ.get("HS_SYNTH_CACHE", "/tmp/hs_synth"))
    d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
    bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
    nr = torch.zeros(1, dtype=torch.int32, device=dev)
    for label, kw in (("whole chip", {}), ("one 32-CU slice", {"partition": (0, 8)})):
        ctx = _lib.DeviceContext(n, **kw)
        st = torch.cuda.ExternalStream(ctx.stream, device=dev)
        torch.cuda.synchronize(dev)
        cells = []
        for sel in (_lib.KERNEL_OCT, _lib.KERNEL_OCT_SPLIT, _lib.KERNEL_FUSED, _lib.KERNEL_AUTO):
            ctx.set_kernel(sel)
            ctx.set_profiling(120)
            walls = []
            for i in range(110):
                st.synchronize()
                t0 = time.perf_counter()
                ctx.verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), nr.data_ptr(), ctx.stream)
                st.synchronize()
                if i >= 10:
                    walls.append((time.perf_counter() - t0) * 1e3)
            a, b, k = ctx.kernel_times()
            walls.sort()
            cells.append("%.4f / %.4f" % ((a + b) / max(1, k), walls[len(walls) // 2]))
        assert int(nr.item()) == 0
        ctx.close()
        print("| %d | %s | %s |" % (n, label, " | ".join(cells)), flush=True)
