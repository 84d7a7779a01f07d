"""Per-kernel averages of the counters in a rocprofv3 --pmc results DB (rocpd sqlite), as JSON.
usage: python tools/rocpd_pmc.py <results.db> [kernel-name-substring ...]
A counter that is reported per instance (per XCD / channel) is summed over the instances of a dispatch first."""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
want = sys.argv[2:]
rows = db.execute("select name, dispatch_id, counter_name, sum(counter_value) from pmc_events "
                  "group by name, dispatch_id, counter_name").fetchall()
out = {}
for name, _, counter, value in rows:
    short = name.split("(")[0]
    if want and not any(w in short for w in want):
        continue
    k = out.setdefault(short, {}).setdefault(counter, [0, 0.0])
    k[0] += 1
    k[1] += value
print(json.dumps({k: {c: {"dispatches": v[0], "avg_per_dispatch": v[1] / v[0]} for c, v in cs.items()} for k, cs in out.items()},
                 indent=1))
