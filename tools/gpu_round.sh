#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_tickets.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
S="16x1x1 8x1x1 16x2x2 16x4x2 16x4x4 8x4x2 4x8x2 1x32x2"
for d in 1 2; do for p in 16 8; do
HS_ED25519_POOL_DEPTH=$d HS_ED25519_POOL_PARTITIONS=$p timeout 120 python tools/e2e_async.py $S 2>&1 | grep -v amdgpu.ids | sed "s/^/depth=$d /"
done; done
