#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== full suite"; timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/t_all.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/t_all.log
