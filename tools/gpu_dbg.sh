#!/bin/bash
mkdir -p gpurun_out; export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== qc"; timeout 300 python tools/dbg1.py qc 2>&1 | tail -15
echo "== rlc cold"; timeout 300 python tools/dbg1.py rlc 6667 2>&1 | tail -5
echo "== rlc cached"; timeout 300 python tools/dbg1.py rlc 3000 cached 2>&1 | tail -12
echo "== prof 6667"; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rlc6667 -o rlc -- python tools/dbg1.py rlc 6667 > gpurun_out/prof6667.log 2>&1; tail -3 gpurun_out/prof6667.log
echo "== prof 66667"; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rlc66667 -o rlc -- python tools/dbg1.py rlc 66667 > gpurun_out/prof66667.log 2>&1; tail -3 gpurun_out/prof66667.log
find gpurun_out/prof_rlc6667 gpurun_out/prof_rlc66667 -name "*kernel_stats*" | head
for f in $(find gpurun_out/prof_rlc6667 gpurun_out/prof_rlc66667 -name "*kernel_stats.csv"); do echo $f; head -12 $f | cut -c1-160; done
