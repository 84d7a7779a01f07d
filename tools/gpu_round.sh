#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "## key cache, after the 16-bit base windows" > gpurun_out/e2e_cached2.txt
HS_E2E_CACHED=1 timeout 300 python tools/e2e_async.py 1x1x1 16x1x1 16x2x2 16x4x4 16x4x8 4x8x2 4x8x4 1x32x2 1x32x4 2>/dev/null >> gpurun_out/e2e_cached2.txt
echo "## cold, after the pairs form" >> gpurun_out/e2e_cached2.txt
timeout 300 python tools/e2e_async.py 16x2x2 16x4x4 16x8x4 8x8x4 4x8x2 4x16x4 1x32x2 1x64x4 2>/dev/null >> gpurun_out/e2e_cached2.txt
cat gpurun_out/e2e_cached2.txt
timeout 300 python tools/streamed_10s.py > gpurun_out/r2_streamed_10s.md 2>/dev/null; cat gpurun_out/r2_streamed_10s.md
