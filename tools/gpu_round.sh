#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py tests/test_gpu_rlc.py tests/test_gpu_fuzz_apis.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
timeout 300 python tools/lane_probe.py 66667 262144 2>/dev/null
show() { python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('headline %.2f M  kernel %s %.4f ms (prepare %s)  int_roofline %.3f  K %d partitions %s' % (
        j['value'] / 1e6, j['roofline']['kernel'], j['roofline']['avg_kernel_ms'], j['kernels_ms']['k_prepare'], (j['int_roofline'] or {}).get('frac', 0), j['config']['batches_per_launch'], j['config']['cu_partitions']))
"; }
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-cached --no-qc"
echo "== pairs K=4"; $B 2>/dev/null | show
echo "== lane K=16 plain streams"; $B --kernel lane --batches-per-launch 16 --partitions 0 2>/dev/null | show
echo "== lane K=16 4 partitions"; $B --kernel lane --batches-per-launch 16 --partitions 4 2>/dev/null | show
echo "== lane K=8 plain streams"; $B --kernel lane --batches-per-launch 8 --partitions 0 2>/dev/null | show
echo "== lane K=8 4 partitions"; $B --kernel lane --batches-per-launch 8 --partitions 4 2>/dev/null | show
