#!/bin/bash
# One GPU-box visit: new/risky tests first, then the whole -m gpu suite, then the bench as the driver runs it.
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== new tests"; timeout 1200 python -m pytest tests/test_gpu_rlc.py tests/test_gpu_sharded.py tests/test_gpu_baseline_sizes.py -x -q -m gpu > gpurun_out/t_new.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/t_new.log
echo "== full suite"; timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/t_all.log
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "rc=$?"; tail -3 gpurun_out/bench.err; head -c 600 gpurun_out/bench.json
for n in 6667 66667; do
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rlc$n -o rlc -- python tools/dbg1.py rlc $n > gpurun_out/prof$n.log 2>&1; grep "^mode" gpurun_out/prof$n.log
  python tools/rocpd_summary.py gpurun_out/prof_rlc$n/rlc_results.db > gpurun_out/rlc_${n}_kernel_stats.csv 2>/dev/null; head -12 gpurun_out/rlc_${n}_kernel_stats.csv | cut -c1-60,140-
done
