#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_all.txt 2>&1; echo rc=$?
grep -E "passed|failed|error" gpurun_out/pytest_gpu_all.txt | tail -5; grep -n -E "^(FAILED|ERROR)|Error|assert " gpurun_out/pytest_gpu_all.txt | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
