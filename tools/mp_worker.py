"""One node PROCESS of bench.py's `multi_process` extra and of tests/test_gpu_multiprocess.py: its own engine
instance on device 0 (tenancy from the environment), the end-to-end host-buffer load from native threads.

    python tools/mp_worker.py <signatures per call> <seconds> <index> [threads inflight burst]

Prints "ready" once initialised, waits for a line on stdin (so that all workers start together), runs, prints one
JSON object."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    n, seconds, index = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
    threads, inflight, burst = (int(x) for x in sys.argv[4:7]) if len(sys.argv) >= 7 else (4, 8, 4)
    try:
        import synth
        from hotstuff_amd import _lib
        _lib.init(0, n)
        nb = 4
        batches = [synth.make_batch(n, shared=False, first=(index * nb + j) * n, cache_dir="/tmp/hs_synth") for j in range(nb)]
        dg, pk, sg = (b"".join(b[i] for b in batches) for i in range(3))
        bufs = [ctypes.create_string_buffer(x, len(x)) for x in (dg, pk, sg)]
        a = [ctypes.addressof(x) for x in bufs]
        _lib.loadgen(None, a[0], 32, a[1], a[2], n, nb, _lib.STRICT, threads, inflight, burst, 0.1)        # warm: pools, workspaces
    except Exception as e:
        print("ready (failed)", flush=True)
        sys.stdin.readline()
        print(json.dumps({"error": "%s: %s" % (type(e).__name__, e)}), flush=True)
        return
    print("ready", flush=True)
    sys.stdin.readline()
    try:
        calls, p50, el = _lib.loadgen(None, a[0], 32, a[1], a[2], n, nb, _lib.STRICT, threads, inflight, burst, seconds)
        print(json.dumps({"verifies_per_s": calls * n / el, "p50_ms": p50, "calls": calls, "verdict_errors": 0}), flush=True)
    except Exception as e:
        print(json.dumps({"error": "%s: %s" % (type(e).__name__, e)}), flush=True)


if __name__ == "__main__":
    main()
