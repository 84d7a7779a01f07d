"""Per-queue busy fraction and per-kernel average durations from a rocprofv3 --kernel-trace results DB (rocpd sqlite):
how much of the time between a queue's first and last dispatch it was running a kernel, and which kernels.
usage: python tools/rocpd_streams.py <results.db> [t_from_fraction t_to_fraction]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.3, 0.9)
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels order by start").fetchall() if "queue_id" in cols else None
if rows is None:
    print("columns of view `kernels`:", cols)
    sys.exit(1)
t_lo, t_hi = rows[0][1], max(r[2] for r in rows)
a, b = t_lo + f0 * (t_hi - t_lo), t_lo + f1 * (t_hi - t_lo)
rows = [r for r in rows if r[1] >= a and r[2] <= b]
per_q, per_k, names = {}, {}, {}
for name, s, e, q, gx, wx in rows:
    per_q.setdefault(q, []).append((s, e))
    names.setdefault(q, []).append(name.split("(")[0])
    k = per_k.setdefault(name.split("(")[0], [0, 0.0])
    k[0] += 1; k[1] += (e - s)
print("window %.1f ms, %d dispatches on %d queues" % ((b - a) / 1e6, len(rows), len(per_q)))
for name, (cnt, tot) in sorted(per_k.items(), key=lambda kv: -kv[1][1]):
    print("  %-40s x %6d  avg %.4f ms" % (name, cnt, tot / cnt / 1e6))
for q, iv in sorted(per_q.items()):
    busy = sum(e - s for s, e in iv)
    span = iv[-1][1] - iv[0][0]
    gaps = sorted((iv[i + 1][0] - iv[i][1]) / 1e3 for i in range(len(iv) - 1))
    print("  queue %s: %4d kernels, busy %.3f of its span; gap between consecutive kernels p50 %.1f us p90 %.1f us" % (
        q, len(iv), busy / max(1, span), gaps[len(gaps) // 2] if gaps else 0, gaps[int(len(gaps) * 0.9)] if gaps else 0))
by_pair = {}
for q, iv in per_q.items():
    for i in range(len(iv) - 1):
        by_pair.setdefault((names[q][i], names[q][i + 1]), []).append((iv[i + 1][0] - iv[i][1]) / 1e3)
print("gap between consecutive kernels of one queue, by (previous -> next):")
for (a_, b_), g in sorted(by_pair.items(), key=lambda kv: -len(kv[1])):
    g.sort()
    print("  %-34s -> %-34s x %5d  p10 %.1f  p50 %.1f  p90 %.1f us" % (a_, b_, len(g), g[len(g) // 10], g[len(g) // 2], g[int(len(g) * 0.9)]))
