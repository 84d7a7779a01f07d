"""One synchronous 1,024-signature strict host call (what an unmodified Core issues: hs_ed25519_verify_multi, narrow pool, a 32-CU slice)
beside the GPU's measured eight-lane pick - the row the narrow-context fix moved on GPUs whose pick is the split form.
    python tools/one_sync_probe.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import synth
from hotstuff_amd import _lib
_lib.init(0, 8192)
d = _lib.config_dump()["devices"][0]
n = 1024
dg, pk, sg = synth.make_batch(n, shared=False, cache_dir="/tmp/hs_synth")
b = [ctypes.create_string_buffer(x, len(x)) for x in (dg, pk, sg)]
lib = _lib.lib()
ts = []
for i in range(320):
    t0 = time.perf_counter()
    rc = lib.hs_ed25519_verify_multi(ctypes.addressof(b[0]), ctypes.addressof(b[1]), ctypes.addressof(b[2]), n, _lib.STRICT, None)
    dt = time.perf_counter() - t0
    assert rc == 1
    if i >= 20: ts.append(dt * 1e3)
ts.sort()
print("pick", d["oct_pick"], d["oct_ms"], "| one synchronous 1,024-signature strict host call p50 %.4f ms p90 %.4f" % (ts[len(ts)//2], ts[(9*len(ts))//10]))
