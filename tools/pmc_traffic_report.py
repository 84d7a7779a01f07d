"""Merge the solo-launch counter passes (tools/pmc_solo.py under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, summarised by
tools/rocpd_pmc.py --by-grid) with the calibration (tools/pmc_calibrate.py) into profiles/r3_pmc_traffic.json: per kernel
and launch size, the counters as reported, the calibration factor applied and the corrected HBM-side bytes per launch.

    python tools/pmc_traffic_report.py <solo FETCH json> <solo WRITE json> <calibration json> > profiles/r3_pmc_traffic.json
"""
import json
import sys

fetch, write, cal = (json.load(open(p)) for p in sys.argv[1:4])
F = cal["factors"]
# which calibrated pattern dominates each kernel's reads (every read pattern measured reports half the touched
# 128-byte lines, so the choice moves the result by < 5 %) and what it moves by design per signature
KERNELS = {
    "k_verify_batches@256": ("headline pair, 16 x 1,024 signatures per launch", 16384, "fetch_line_gather_16B", 128 * 16384),
    "k_prepare_batches@512": ("headline pair, 16 x 1,024 signatures per launch", 16384, "fetch_dword_rows", 128 * 16384),
    "k_verify_fused_pairs_batches@128": ("pairs form, 4 x 1,024 signatures per launch", 4096, "fetch_line_gather_16B", 128 * 4096),
    "k_verify_fused_batches@128": ("one signature per quad, 2 x 1,024 signatures per launch", 2048, "fetch_line_gather_16B", 128 * 2048),
    "k_verify_cached_lane2r_batches@64": ("key cache, 4 x 1,024 signatures per launch", 4096, "fetch_line_gather_16B", (48 * 128 + 100) * 4096),
    "k_verify_cached_lane2r@1042": ("key cache, one 66,667-signature QC (N = 100,000)", 66667, "fetch_line_gather_16B", (48 * 128 + 68) * 66667),
    "k_prepare@2084": ("cold one-lane pair, 66,667 signatures", 66667, "fetch_dword_rows", 96 * 66667),
    "k_verify_tail@1096": ("cold stage 2 of one 66,667-signature QC: 1,024 one-lane wavefronts + the 1,131-signature sliver as 72 four-lane wavefronts of the same grid", 66667, "fetch_line_gather_16B", 96 * 66667),
    "k_verify@1024": ("(earlier form) cold one-lane pair, 65,536 of 66,667 signatures (the sliver of 1,131 ran through k_verify_fused@72 on a side stream)", 65536, "fetch_line_gather_16B", 96 * 65536),
    "k_prepare@2048": ("cold one-lane pair, 65,536 of 66,667 signatures", 65536, "fetch_dword_rows", 96 * 65536),
    "k_verify_fused@72": ("the 1,131-signature sliver of the 66,667-signature call", 1131, "fetch_line_gather_16B", 96 * 1131),
    "k_rlc_prepare@3126": ("RLC stage 1, 66,667 signatures", 66667, "fetch_dword_rows", 96 * 66667),
    "k_rlc_accum@133": ("RLC stage 3, 66,667 signatures: 48 record gathers per signature", 66667, "fetch_line_gather_16B", 48 * 132 * 66667),
    "k_rlc_scatter@8": ("RLC stage 2, 66,667 signatures", 66667, "fetch_dword_rows", 0),
    "k_rlc_hist@8": ("RLC stage 2, 66,667 signatures", 66667, "fetch_dword_rows", 0),
    "k_rlc_reduce@33": ("RLC stage 4", 66667, "fetch_line_gather_16B", 0),
    "k_rlc_final@1": ("RLC stage 5", 66667, "fetch_line_gather_16B", 0),
}
# the 160-byte-entry gather reads two 128-byte lines per entry; its factor against TOUCHED LINES is what transfers
lines160 = cal["kernels"]["k_cal_gather_entry160"].get("fetch_factor_vs_touched_line_bytes_128")
factor = dict(F, fetch_entry160_gather_lines=lines160)
out = {"note": "SOLO launches (one dispatch at a time, the stream drained in between: tools/pmc_solo.py), rocprofv3 --pmc FETCH_SIZE and "
               "--pmc WRITE_SIZE in separate passes, no trace domains; counters summed over their instances, averaged over the dispatches of "
               "a kernel at one grid size.  CORRECTED with this project's own calibration (profiles/pmc_calibration.json, "
               "tools/ubench_gather.hip): FETCH_SIZE reports HALF of the 128-byte lines fetched in EVERY read pattern measured (wide 16-byte "
               "streams 2.00, dword rows 2.00, one-line 16-byte gathers 1.91, dword gathers 2.00 against touched lines), so reads are "
               "multiplied by the factor of the kernel's dominant pattern; WRITE_SIZE is exact for full-line writes (1.00) and is taken as "
               "reported - for 16-byte pieces of a line it over-reports once the footprint exceeds the caches (factor 0.43 on a 4 GiB "
               "footprint), which none of these kernels' write sets does.  Infinity-Cache hits are counted by these counters "
               "(MI355X_MICROARCH.md), so 'HBM-side' means 'left the L2'.",
       "calibration": "profiles/pmc_calibration.json", "factors": factor, "kernels": {}}
for k, (what, sigs, pat, algo) in KERNELS.items():
    if k not in fetch:
        continue
    f_kb = fetch[k]["FETCH_SIZE"]["avg_per_dispatch"]
    w_kb = write.get(k, {}).get("WRITE_SIZE", {}).get("avg_per_dispatch", 0.0)
    fac = factor[pat]
    rd, wr = f_kb * 1024 * fac, w_kb * 1024
    row = {"what": what, "signatures_per_launch": sigs, "dispatches": fetch[k]["FETCH_SIZE"]["dispatches"],
           "FETCH_SIZE_KB_reported": round(f_kb, 1), "WRITE_SIZE_KB_reported": round(w_kb, 1),
           "read_pattern": pat, "fetch_factor": fac,
           "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
           "hbm_bytes_per_signature": round((rd + wr) / sigs, 1)}
    if algo:
        row["algorithmic_bytes_per_launch"] = algo
        row["traffic_over_algorithmic"] = round((rd + wr) / algo, 2)
    out["kernels"][k.split("@")[0] + "@" + k.split("@")[1]] = row
print(json.dumps(out, indent=1))
