"""What does hs_ed25519_autotune measure, against bench.py's own A/B of the two eight-lane cold forms?  (round 5: the first form of the
autotune alternated single runs of the two forms and read 0.39 / 0.42 ms where blocks of one form read 0.25 / 0.30 ms - every run
started on the other kernel's instructions in the caches.)  python tools/tune_probe.py  -> profiles/r5_c_tune_probe.txt"""
import os, sys, time, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, synth
from hotstuff_amd import _lib
t0 = time.time()
_lib.init(0, 8192)
print("init took %.1f ms" % ((time.time() - t0) * 1e3), _lib.config_dump()["devices"][0]["oct_ms"], flush=True)
for i in range(4):
    t0 = time.time(); r = _lib.autotune(0); print("autotune again (%.1f ms):" % ((time.time() - t0) * 1e3), r, flush=True)
dev = torch.device("cuda", 0)
n = 6667
def ab(label, dg, pk, sg, reps=3):
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
    bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
    ctx = _lib.DeviceContext(n)
    st = torch.cuda.ExternalStream(ctx.stream, device=dev)
    torch.cuda.synchronize(dev)
    for rep in range(reps):
        out = {}
        for name, sel in (("fused8", _lib.KERNEL_OCT), ("fused8h", _lib.KERNEL_OCT_SPLIT)):
            ctx.set_kernel(sel); ctx.set_profiling(10)
            for i in range(10):
                ctx.verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), 0, 0)
                st.synchronize()
            k = ctx.kernel_times(); out[name] = round((k[0] + k[1]) / k[2], 4)
        print(label, rep, out, flush=True)
    ctx.close()
dg, pk, sg = synth.make_batch(n, shared=True, cache_dir="/tmp/hs_synth")
ab("distinct keys", dg, pk, sg)
# 64 keys repeated, signed on the GPU (what the autotune's canned batch is)
seeds = [synth.key_seed(900_000 + i) for i in range(64)]
pks = _lib.public_keys(b"".join(seeds), 64)
sec = b"".join(seeds[i % 64] + pks[32 * (i % 64):32 * (i % 64) + 32] for i in range(n))
pk64 = b"".join(pks[32 * (i % 64):32 * (i % 64) + 32] for i in range(n))
sg64 = _lib.sign_batch(sec, dg, n, shared=True)
ab("64 keys", dg, pk64, sg64)
t0 = time.time(); r = _lib.autotune(0); print("autotune after load (%.1f ms):" % ((time.time() - t0) * 1e3), r, flush=True)
time.sleep(2.0)
t0 = time.time(); r = _lib.autotune(0); print("autotune after 2 s idle (%.1f ms):" % ((time.time() - t0) * 1e3), r, flush=True)
