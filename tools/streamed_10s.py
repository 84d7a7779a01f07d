"""BASELINE configs[4] as BASELINE.md section 4 words it: back-to-back 6,667-signature QC rounds for >= 10 s from ONE host
thread through the pinned round ring (4 slots, the engine's AUTO choice = zero-copy at this size), cold and with the
committee registered; sustained signatures/s and submit -> verdict p50 / p99.   python tools/streamed_10s.py [seconds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402
from hotstuff_amd import _lib  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
n, slots = 6667, 4
dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=os.environ.get("HS_SYNTH_CACHE", "/tmp/hs_synth"))
keys = [pk[32 * i:32 * i + 32] for i in range(n)]
print("| path | seconds | rounds | M signatures/s | submit->verdict p50 ms | p99 ms | max ms |")
print("|---|---|---|---|---|---|---|")
for cached in (False, True):
    ks = _lib.KeySet(keys) if cached else None
    kb = ks.indices(keys)[0] if cached else pk
    src = [np.frombuffer(x, dtype=np.uint8) for x in (dg, kb, sg)]
    ring = _lib.RoundRing(n, slots=slots, keyset=ks, kind="auto")
    inflight, lat, t_sub, done, r = set(), [], {}, 0, 0
    t0 = time.perf_counter()
    t_end = t0 + seconds
    while time.perf_counter() < t_end or inflight:
        slot = r % slots
        if slot in inflight:
            ok, _ = ring.wait(slot, want_bitmap=False)
            lat.append(time.perf_counter() - t_sub[slot])
            assert ok, "a streamed round was rejected"
            inflight.discard(slot)
            done += 1
        if time.perf_counter() < t_end:
            d, k, s_ = ring.begin(slot, n, True)
            d[:] = src[0]; k[:] = src[1]; s_[:] = src[2]
            t_sub[slot] = time.perf_counter()
            ring.submit(slot, _lib.BATCH)
            inflight.add(slot)
        r += 1
    wall = time.perf_counter() - t0
    ring.close()
    if ks:
        ks.close()
    lat.sort()
    print("| %s | %.1f | %d | %.1f | %.3f | %.3f | %.3f |" % ("key cache" if cached else "cold", wall, done, done * n / wall / 1e6,
                                                               lat[len(lat) // 2] * 1e3, lat[int(len(lat) * 0.99)] * 1e3, lat[-1] * 1e3), flush=True)
