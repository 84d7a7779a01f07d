"""Dump the per-kernel statistics of a rocprofv3 (rocpd sqlite) results DB as CSV.
usage: python tools/rocpd_summary.py <results.db> > profiles/<name>.csv"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
w = csv.writer(sys.stdout)
w.writerow(["name", "calls", "total_us", "avg_us", "percent"])
for r in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    w.writerow([r[0], r[1], "%.3f" % r[2], "%.3f" % r[3], "%.4f" % r[4]])
