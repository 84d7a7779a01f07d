"""Inputs brought in coalesced through LDS (k_prepare_batches_staged, k_verify_cached_lane2r[_batches]) against the per-lane
record loads they replace (HS_ED25519_PREPARE_STAGED=0 / HS_ED25519_CACHED_STAGED=0 select the old kernels of the same
library), same box, alternating: host-API rates from pinned staging read over PCIe (tools/e2e_async.py), the key-cache QC
latency, and the HBM-resident bench rows.  The two switches exist at commit 8f5d7a8 only (profiles/r4_f_staged_ab.md is its
output); since then the winners are built in: staged batch lists, per-lane loads for the lone key-cache call.

    python tools/staged_ab.py [quick]      (driver; every measurement runs in a subprocess of its own)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
CONFIGS = (("staged (new)", {}), ("per-lane loads (old)", {"HS_ED25519_PREPARE_STAGED": "0", "HS_ED25519_CACHED_STAGED": "0"}))
COLD_SHAPES = ["16x16x16", "4x32x16", "1x256x16", "16x8x8", "16x4x4"]
CACHED_SHAPES = ["16x16x16", "1x256x16", "16x8x8", "16x4x4"]


def qc_worker():
    """key-cache QC of 6,667 signatures: device-resident p50 + kernel time, and the host API from pageable buffers"""
    import torch
    import synth
    from hotstuff_amd import _lib
    dev = torch.device("cuda:0")
    _lib.init(0, 8192)
    n = 6667
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir="/tmp/hs_synth")
    ks = _lib.KeySet([pk[32 * i:32 * i + 32] for i in range(n)])
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_dg, d_sg = t(dg), t(sg)
    d_ki = torch.arange(n, dtype=torch.int32, device=dev)
    bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
    nr = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx = _lib.DeviceContext(n)
    ctx.set_profiling(256)
    st = torch.cuda.ExternalStream(ctx.stream, device=dev)
    torch.cuda.synchronize(dev)
    ts = []
    with torch.cuda.stream(st):
        for i in range(220):
            nr.zero_(); st.synchronize()
            t0 = time.perf_counter()
            ctx.verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), nr.data_ptr(), ctx.stream)
            bad = int(nr.item())
            dt = time.perf_counter() - t0
            assert bad == 0
            if i >= 20:
                ts.append(dt)
    ts.sort()
    a, b, k = ctx.kernel_times()
    out = {"device_p50_ms": round(ts[len(ts) // 2] * 1e3, 4), "kernel_ms": round(b / max(1, k), 4)}
    ctx.close()
    # the host API (what the Rust shim binds): pageable host buffers -> pinned staging -> zero-copy kernel -> host verdict
    import struct
    idx = struct.pack("<%dI" % n, *range(n))
    hs = []
    for i in range(220):
        t0 = time.perf_counter()
        ok, _ = ks.verify_packed(dg, idx, sg, n, _lib.BATCH, True)
        dt = time.perf_counter() - t0
        assert ok
        if i >= 20:
            hs.append(dt)
    if hs:
        hs.sort()
        out["host_api_p50_ms"] = round(hs[len(hs) // 2] * 1e3, 4)
    ks.close()
    print(json.dumps(out))


def run(cmd, env, timeout=600):
    o = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    if o.returncode != 0:
        return None, (o.stderr or o.stdout)[-600:]
    return o.stdout, None


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "qc_worker":
        return qc_worker()
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    py = sys.executable
    print("# staged (coalesced, LDS) input loads against per-lane record loads - same box, alternating\n")
    rounds = 1 if quick else 2
    for r in range(rounds):
        for name, env in CONFIGS:
            out, err = run([py, "tools/e2e_async.py"] + COLD_SHAPES, env)
            print("## host API, cold, %s (pass %d)\n%s" % (name, r + 1, out if out else "FAILED: " + err), flush=True)
            out, err = run([py, "tools/e2e_async.py"] + CACHED_SHAPES, dict(env, HS_E2E_CACHED="1"))
            print("## host API, key cache, %s (pass %d)\n%s" % (name, r + 1, out if out else "FAILED: " + err), flush=True)
            out, err = run([py, os.path.abspath(__file__), "qc_worker"], env)
            print("## key-cache QC of 6,667 signatures, %s (pass %d)\n%s" % (name, r + 1, out if out else "FAILED: " + err), flush=True)
    for name, env in CONFIGS:
        out, err = run([py, "bench.py", "--steps", "10", "--warmup", "3", "--no-cpu", "--no-qc"], env)
        if not out:
            print("## bench, %s\nFAILED: %s" % (name, err), flush=True)
            continue
        j = json.loads(out.strip().splitlines()[-1])
        cc = j.get("committee_cached") or {}
        print("## bench (HBM-resident), %s\nheadline %.2f M verifies/s  k_prepare %.4f ms  k_verify %.4f ms | committee_cached %.1f M (kernel %.4f ms), "
              "large committee kernel %s ms | end_to_end %s\n" % (
                  name, j["value"] / 1e6, j["kernels_ms"]["k_prepare"] or 0, j["kernels_ms"]["k_verify"], (cc.get("value") or 0) / 1e6,
                  cc.get("avg_kernel_ms") or 0, (cc.get("large_committee") or {}).get("avg_kernel_ms"),
                  [(r_["host_threads"], r_["in_flight_per_thread"], r_["batches_per_launch"], round(r_["value"] / 1e6, 1))
                   for r_ in (j.get("end_to_end") or {}).get("by_threads", [])]), flush=True)


if __name__ == "__main__":
    main()
