"""Combine tools/ubench_gather's known volumes with rocprofv3's FETCH_SIZE / WRITE_SIZE for the same run into the
calibration factors bench.py and the profiles use.

    python tools/pmc_calibrate.py <ubench stdout json> <FETCH_SIZE results.db> <WRITE_SIZE results.db> > profiles/pmc_calibration.json

factor = true bytes / counter bytes (multiply a counter reading by it).  rocprofv3 reports the counters in KB."""
import json
import sqlite3
import sys


def per_kernel(dbpath, counter):
    db = sqlite3.connect(dbpath)
    rows = db.execute("select name, dispatch_id, sum(counter_value) from pmc_events where counter_name = ? group by name, dispatch_id",
                      (counter,)).fetchall()
    out = {}
    for name, did, v in rows:
        out.setdefault(name.split("(")[0], []).append(v)
    # the LAST dispatch of each kernel (the first also pays cold page tables)
    return {k: v[-1] * 1024.0 for k, v in out.items()}


known = json.load(open(sys.argv[1]))["kernels"]
fetch = per_kernel(sys.argv[2], "FETCH_SIZE")
write = per_kernel(sys.argv[3], "WRITE_SIZE")
res = {"note": "factor = true bytes / counter bytes; multiply a rocprofv3 reading by it.  4 GiB buffers (beyond L2 and the Infinity "
               "Cache), every byte touched once; the second of two dispatches.  tools/ubench_gather.hip, tools/pmc_calibrate.py",
       "kernels": {}}
for k, info in known.items():
    row = dict(info)
    if "read_bytes" in info and k in fetch:
        row["FETCH_SIZE_bytes"] = fetch[k]
        row["fetch_factor_vs_useful_bytes"] = round(info["read_bytes"] / fetch[k], 4)
        for g in ("touched_line_bytes_128", "touched_line_bytes_64"):
            if g in info:
                row["fetch_factor_vs_" + g] = round(info[g] / fetch[k], 4)
    if "write_bytes" in info and k in write:
        row["WRITE_SIZE_bytes"] = write[k]
        row["write_factor"] = round(info["write_bytes"] / write[k], 4)
    res["kernels"][k] = row
g = res["kernels"]
res["factors"] = {
    "fetch_wide_16B_coalesced": g.get("k_cal_stream_read16", {}).get("fetch_factor_vs_useful_bytes"),
    "fetch_dword_rows": g.get("k_cal_stream_read4", {}).get("fetch_factor_vs_useful_bytes"),
    "fetch_line_gather_16B": g.get("k_cal_gather_line16", {}).get("fetch_factor_vs_useful_bytes"),
    "fetch_entry160_gather": g.get("k_cal_gather_entry160", {}).get("fetch_factor_vs_useful_bytes"),
    "fetch_dword_gather_vs_128B_lines": g.get("k_cal_gather_dword", {}).get("fetch_factor_vs_touched_line_bytes_128"),
    "write_wide_16B": g.get("k_cal_stream_write16", {}).get("write_factor"),
    "write_dword_rows": g.get("k_cal_row_write4", {}).get("write_factor"),
    "write_entry_16B_pieces": g.get("k_cal_entry_write16", {}).get("write_factor"),
}
print(json.dumps(res, indent=1))
