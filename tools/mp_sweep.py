"""Four node processes on one GPU (BASELINE configs[0]'s topology, /root/reference/benchmark/benchmark/local.py:94-104):
which submission shape and which split of the chip gets them to the single-process rate (VERDICT r3 "Next round" #2: >= 65 M
verifies/s from four tenants, max / min per process <= 1.5)?  Each row = k processes of tools/mp_worker.py started together.

    python tools/mp_sweep.py [seconds]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8


def row(k, shape, env, tenancy=True):
    r = bench.multi_process(1024, procs=(k,), seconds=seconds, shape=shape, extra_env=env, tenancy_modes=(tenancy,))["rows"][0]
    pp = r.get("per_process", [])
    print("%d proc  threads x tickets / burst %-12s tenancy %-5s %-58s -> %6.1f M  per process %s  max/min %.2f  p50 %.3f ms  %s" % (
        k, "%dx%d/%d" % shape, tenancy, " ".join("%s=%s" % (a.replace("HS_ED25519_", ""), b) for a, b in sorted(env.items())),
        r.get("value", 0) / 1e6, [round(x / 1e6, 1) for x in pp], (max(pp) / max(min(pp), 1)) if pp else 0, r.get("call_p50_ms", 0),
        r.get("error", "")), flush=True)
    return r


row(1, (4, 8, 4), {}, False)
row(1, (4, 32, 16), {}, False)
# the default tenant pools (round 6: 4 narrow streams x 2 contexts on the tenant's ONE 64-CU slice) at bursts of four and sixteen, then
# round 5's split (two 32-CU slices) and queue depth 1, at both burst sizes
SPLITS = ({}, {"HS_ED25519_POOL_PARTITIONS": "2"}, {"HS_ED25519_POOL_DEPTH": "1"}, {"HS_ED25519_POOL_PARTITIONS": "2", "HS_ED25519_POOL_DEPTH": "1"})
for env in SPLITS:
    for shape in ((4, 8, 4), (4, 32, 16), (1, 128, 16), (2, 64, 16)):
        row(4, shape, env)
row(4, (4, 32, 16), {}, False)
row(2, (4, 8, 4), {})
row(2, (4, 64, 16), {})
