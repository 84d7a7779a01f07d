#!/bin/bash
# Round-2 evidence: the bench as the driver runs it, rocprofv3 kernel stats of the same command, and the HBM
# traffic counters in separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass: TCC slots).
mkdir -p gpurun_out/r2
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2
echo "== plain bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/r2_bench_default.json.txt 2> $O/bench.err; echo rc=$?; head -c 200 $O/r2_bench_default.json.txt; echo
echo "== kernel trace"; timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu > $O/r2_bench_under_rocprof.json.txt 2> $O/kt.err; echo rc=$?
python tools/rocpd_summary.py $O/kt/kt_results.db > $O/r2_bench_default_kernel_stats.csv; head -8 $O/r2_bench_default_kernel_stats.csv | cut -c1-50,150-
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c (headline)"; timeout 600 rocprofv3 --pmc $c -d $O/pmc_$c -o p -- python bench.py --steps 3 --warmup 1 --no-extras --no-cpu > /dev/null 2> $O/pmc_$c.err; echo rc=$?
  python tools/rocpd_pmc.py $O/pmc_$c/p_results.db k_verify > $O/r2_pmc_${c}_headline.json; cat $O/r2_pmc_${c}_headline.json | tr -d '\n '; echo
  echo "== pmc $c (rlc 66667)"; timeout 600 rocprofv3 --pmc $c -d $O/pmcr_$c -o p -- python tools/rlc_probe.py rlc 66667 > /dev/null 2> $O/pmcr_$c.err; echo rc=$?
  python tools/rocpd_pmc.py $O/pmcr_$c/p_results.db k_rlc k_verify k_prepare > $O/r2_pmc_${c}_rlc66667.json; cat $O/r2_pmc_${c}_rlc66667.json | tr -d '\n '; echo
done
echo "== SQ counters (headline)"; timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $O/pmc_sq -o p -- python bench.py --steps 3 --warmup 1 --no-extras --no-cpu > /dev/null 2> $O/pmc_sq.err; echo rc=$?
python tools/rocpd_pmc.py $O/pmc_sq/p_results.db k_verify > $O/r2_pmc_sq_headline.json; cat $O/r2_pmc_sq_headline.json | tr -d '\n '; echo
rm -rf $O/kt $O/pmc_* $O/pmcr_*
