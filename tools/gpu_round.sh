#!/bin/bash
# One GPU-box visit: the whole -m gpu suite, the bench as the driver runs it, A/B of batches per launch, RLC kernel profile.
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== full suite"; timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/t_all.log
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "rc=$?"; tail -3 gpurun_out/bench.err; head -c 400 gpurun_out/bench.json; echo
for k in 1 2 8; do echo "== bpl $k"; timeout 300 python bench.py --steps 10 --warmup 3 --batches-per-launch $k --no-extras --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_kernel_ms'], d['int_roofline']['frac'])"; done
for n in 6667 66667; do
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rlc$n -o rlc -- python tools/dbg1.py rlc $n > gpurun_out/prof$n.log 2>&1; grep "^mode" gpurun_out/prof$n.log
  python tools/rocpd_summary.py gpurun_out/prof_rlc$n/rlc_results.db > gpurun_out/rlc_${n}_kernel_stats.csv 2>/dev/null
done
