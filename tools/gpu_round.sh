#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python tools/r2_sweeps.py > gpurun_out/r2_sweeps.md 2> gpurun_out/r2_sweeps.err; echo rc=$?; tail -3 gpurun_out/r2_sweeps.err
cat gpurun_out/r2_sweeps.md
