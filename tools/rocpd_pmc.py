"""Per-kernel averages of the counters in a rocprofv3 --pmc results DB (rocpd sqlite), as JSON.
usage: python tools/rocpd_pmc.py <results.db> [--by-grid] [kernel-name-substring ...]
A counter that is reported per instance (per XCD / channel) is summed over the instances of a dispatch first.
--by-grid keeps launches of different grid sizes apart (key "name@workgroups")."""
import json
import sqlite3
import sys

args = sys.argv[1:]
by_grid = "--by-grid" in args
args = [a for a in args if a != "--by-grid"]
db = sqlite3.connect(args[0])
want = args[1:]
grid = {}
if by_grid:
    for did, gx, wx in db.execute("select dispatch_id, grid_size_x, workgroup_size_x from rocpd_kernel_dispatch"):
        grid[did] = gx // max(1, wx)
rows = db.execute("select name, dispatch_id, counter_name, sum(counter_value) from pmc_events "
                  "group by name, dispatch_id, counter_name").fetchall()
out = {}
for name, did, counter, value in rows:
    short = name.split("(")[0]
    if want and not any(w in short for w in want):
        continue
    if by_grid:
        short = "%s@%d" % (short, grid.get(did, 0))
    k = out.setdefault(short, {}).setdefault(counter, [0, 0.0])
    k[0] += 1
    k[1] += value
print(json.dumps({k: {c: {"dispatches": v[0], "avg_per_dispatch": v[1] / v[0]} for c, v in cs.items()} for k, cs in out.items()},
                 indent=1))
