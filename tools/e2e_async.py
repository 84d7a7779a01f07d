"""End-to-end host-API rate (tools/libhs_loadgen.so) over threads x tickets in flight, for the pool shape given by
HS_ED25519_POOL_NARROW / HS_ED25519_POOL_PARTITIONS in the environment.  One line per shape."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth  # noqa: E402
from hotstuff_amd import _lib  # noqa: E402

n, nb = 1024, 16
# HS_E2E_SLOTS=k: k device slots on GPU 0 (virtual devices: each slot its own base-point tables and pools; cold calls go round the slots)
slots = int(os.environ.get("HS_E2E_SLOTS", "1"))
if slots > 1:
    _lib.init_devices([0] * slots, 8192)
else:
    _lib.init(0, 8192)
lib = _lib.lib()
batches = [synth.make_batch(n, shared=False, first=n * j, cache_dir="/tmp/hs_synth") for j in range(nb)]
import struct  # noqa: E402
cached = os.environ.get("HS_E2E_CACHED") == "1"      # the same batches through a key set holding their 16,384 signers
bufs = [ctypes.create_string_buffer(x, len(x)) for x in (b"".join(b[k] for b in batches) for k in range(3))]
bufs.append(ctypes.create_string_buffer(struct.pack("<%dI" % (n * nb), *range(n * nb)), 4 * n * nb))
a = [ctypes.addressof(x) for x in bufs]
ks = _lib.KeySet([b[1][32 * i:32 * i + 32] for b in batches for i in range(n)]) if cached else None
# threads x tickets in flight per thread x batches per launch (HS_ED25519_SUBMIT_MORE bursts)
# a shape ending in "p": the same from caller-owned pinned buffers (hs_ed25519_host_alloc: read in place, no staging copy)
pin = [_lib.PinnedBuffer(x.raw) for x in bufs[:3]]
pa = [x.addr for x in pin]
shapes = [tuple(int(v) for v in s.rstrip("p").split("x")) + (s.endswith("p"),) for s in (sys.argv[1:] or ["16x1x1", "16x2x1", "16x2x2", "16x4x2", "16x4x4", "8x4x2", "8x8x2", "8x8x4",
                                                                          "4x8x2", "4x16x4", "1x8x2", "1x16x2", "1x32x2", "1x32x4"])]
cfg = _lib.config_dump()
tag = ("key-cache " if cached else "") + ("slots=%d " % slots if slots > 1 else "") + "pool=%s depth=%s partitions=%s" % (cfg["pool_narrow"], cfg.get("pool_narrow_depth", 1), cfg["pool_partitions"])
for th, k, bpl, pinned in shapes:
    src = pa if pinned and not ks else a
    cv, pv, ev = _lib.loadgen(ks.handle if ks else None, src[0], 32, a[3] if ks else src[1], src[2], n, nb, _lib.STRICT, th, k, bpl, 0.5)

    class _V:          # keep the print below as it was
        def __init__(self, v): self.value = v
    calls, p50, el = _V(cv), _V(pv), _V(ev)
    print("%s threads=%d in_flight=%d per_launch=%d%s  %.1f M verifies/s  p50 %.3f ms" % (tag, th, k, bpl, " caller-pinned" if pinned else "", calls.value * n / el.value / 1e6, p50.value), flush=True)
