"""Streamed-rounds sweep for the round ring (hs_ed25519_ring_*): hipGraph launch vs the same
three stream operations issued one by one vs zero-copy kernels, cold and key-cache paths, by pipeline depth.

  python tools/ring_sweep.py > gpurun_out/ring_sweep.md        (on a GPU box)

Depth 1 is a latency measurement (fill excluded: submit -> verdict); deeper pipelines measure
the sustained rate from ONE host thread, pinned-slot fill included.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402
from hotstuff_amd import _lib  # noqa: E402

CACHE = os.environ.get("HS_SYNTH_CACHE", "/tmp/hs_synth")


def run(n, slots, ks, src, kind, seconds=1.0):
    ring = _lib.RoundRing(n, slots=slots, keyset=ks, kind=kind)
    for slot in range(slots):                      # warm: capture + first launch
        d, k, s = ring.begin(slot, n, True)
        d[:] = src[0]; k[:] = src[1]; s[:] = src[2]
        ring.submit(slot, _lib.BATCH)
    for slot in range(slots):
        assert ring.wait(slot, want_bitmap=False)[0]
    inflight, lat, t_sub, done, r = set(), [], {}, 0, 0
    t0 = time.perf_counter()
    t_end = t0 + seconds
    while time.perf_counter() < t_end or inflight:
        slot = r % slots
        if slot in inflight:
            ok, _ = ring.wait(slot, want_bitmap=False)
            lat.append(time.perf_counter() - t_sub[slot])
            assert ok
            inflight.discard(slot)
            done += 1
        if time.perf_counter() < t_end:
            d, k, s = ring.begin(slot, n, True)
            d[:] = src[0]; k[:] = src[1]; s[:] = src[2]
            t_sub[slot] = time.perf_counter()
            ring.submit(slot, _lib.BATCH)
            inflight.add(slot)
        r += 1
    wall = time.perf_counter() - t0
    ring.close()
    lat.sort()
    return done * n / wall, lat[len(lat) // 2] * 1e3, lat[int(len(lat) * 0.99)] * 1e3


def main():
    _lib.init(0, 8192)
    print("| committee N | signatures | path | launch | slots | M signatures/s | submit->verdict p50 ms | p99 ms |")
    print("|---|---|---|---|---|---|---|---|")
    for N in (1000, 10000, 100000):
        n = 2 * N // 3 + 1
        dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
        keys = [pk[32 * i:32 * i + 32] for i in range(n)]
        for cached in (False, True):
            ks = _lib.KeySet(keys) if cached else None
            kb = ks.indices(keys)[0] if cached else pk
            src = [np.frombuffer(x, dtype=np.uint8) for x in (dg, kb, sg)]
            for slots in (1, 2, 4, 8):
                for kind in ("graph", "stream-ops", "zero-copy"):
                    rate, p50, p99 = run(n, slots, ks, src, kind)
                    print("| %d | %d | %s | %s | %d | %.1f | %.3f | %.3f |"
                          % (N, n, "key cache" if cached else "cold", kind,
                             slots, rate / 1e6, p50, p99), flush=True)
            if ks:
                ks.close()


if __name__ == "__main__":
    main()
