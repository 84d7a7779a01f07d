#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -5
S="1x1x1 8x1x1 16x1x1 24x1x1 16x2x1 16x2x2 16x4x2 16x4x4 8x4x2 4x8x2 1x8x2 1x16x2 1x32x2"
{
echo "# hs_ed25519_loadgen: 1,024-signature strict batches, host buffers in, host verdict out (PCIe included), 0.5 s per row"
echo "# threads x tickets in flight per thread x batches per launch; in_flight=1 = the synchronous hs_ed25519_verify_multi"
echo "## default pool: 16 narrow contexts on 8 XCD-aligned partitions of 32 CUs"
timeout 200 python tools/e2e_async.py $S 2>&1 | grep -v amdgpu.ids
echo "## 16 partitions of 16 CUs, one context each (the round-1 / early round-2 shape)"
HS_ED25519_POOL_PARTITIONS=16 timeout 200 python tools/e2e_async.py $S 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r2_e2e_async.txt
cat gpurun_out/r2_e2e_async.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json.txt 2>gpurun_out/bench_default.err; tail -c 400 gpurun_out/bench_default.err; wc -c gpurun_out/bench_default.json.txt
