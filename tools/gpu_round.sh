#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_tickets.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -12
