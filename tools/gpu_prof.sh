#!/bin/bash
# Round-2 evidence: the bench as the driver runs it, rocprofv3 kernel stats of the same command (full and headline only),
# and the HBM traffic counters in separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass: TCC slots).  The RLC
# stages' counters (profiles/r2_pmc_traffic.json "rlc_66667") come from an earlier pass of this script: tools/rlc_probe.py.
mkdir -p gpurun_out/r2
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2
echo "== plain bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/r2_bench_default.json.txt 2> $O/bench.err; echo rc=$?; head -c 200 $O/r2_bench_default.json.txt; echo
echo "== kernel trace, headline only"; timeout 900 rocprofv3 --kernel-trace --stats -d $O/kth -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu --no-extras > $O/r2_bench_headline_under_rocprof.json.txt 2> $O/kth.err; echo rc=$?
python tools/rocpd_summary.py $O/kth/kt_results.db > $O/r2_bench_headline_kernel_stats.csv; head -4 $O/r2_bench_headline_kernel_stats.csv | cut -c1-120
python -c "
import json
for l in open('$O/r2_bench_headline_under_rocprof.json.txt'):
    try: j = json.loads(l)
    except Exception: continue
    print('under rocprof: value %.2f M, events avg_kernel_ms %.4f (prepare %s) over %d launches' % (j['value']/1e6, j['roofline']['avg_kernel_ms'], j['kernels_ms']['k_prepare'], j['roofline']['launches_timed']))
"
echo "== kernel trace, full command"; timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu > $O/r2_bench_under_rocprof.json.txt 2> $O/kt.err; echo rc=$?
python tools/rocpd_summary.py $O/kt/kt_results.db > $O/r2_bench_default_kernel_stats.csv; head -8 $O/r2_bench_default_kernel_stats.csv | cut -c1-50,150-
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c (headline + key cache + large committee, by grid size)"; timeout 600 rocprofv3 --pmc $c -d $O/pmc_$c -o p -- python bench.py --steps 2 --warmup 1 --no-cpu --no-qc > /dev/null 2> $O/pmc_$c.err; echo rc=$?
  python tools/rocpd_pmc.py $O/pmc_$c/p_results.db --by-grid k_verify k_prepare > $O/r2_pmc_${c}_by_grid.json; python -c "
import json,sys
j=json.load(open('$O/r2_pmc_${c}_by_grid.json'))
for k,v in sorted(j.items()):
    for c,x in v.items(): print(k, c, x['dispatches'], round(x['avg_per_dispatch'],1))
"
done
echo "== SQ counters (headline)"; timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $O/pmc_sq -o p -- python bench.py --steps 3 --warmup 1 --no-extras --no-cpu > /dev/null 2> $O/pmc_sq.err; echo rc=$?
python tools/rocpd_pmc.py $O/pmc_sq/p_results.db k_verify k_prepare > $O/r2_pmc_sq_headline.json; cat $O/r2_pmc_sq_headline.json | tr -d '\n '; echo
rm -rf $O/kt $O/kth $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq
echo "== cached batches per launch 8 (4 partitions)"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-qc --cached-batches-per-launch 8 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('cached K=8: %.1f M  (avg kernel %.4f ms)' % (j['committee_cached']['value'] / 1e6, j['committee_cached']['avg_kernel_ms']))"
