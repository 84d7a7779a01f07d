#!/bin/bash
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "several or batches_per_launch" 2>&1 | grep -E "passed|failed|error" | tail -3
show() { python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('headline %.2f M  kernel %s %.4f ms (prepare %s)  int_roofline %.3f  K %d partitions %s' % (
        j['value'] / 1e6, j['roofline']['kernel'], j['roofline']['avg_kernel_ms'], j['kernels_ms']['k_prepare'], (j['int_roofline'] or {}).get('frac', 0), j['config']['batches_per_launch'], j['config']['cu_partitions']))
"; }
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-cached --no-qc"
echo "== K=16"; $B 2>/dev/null | show
echo "== K=32 (one launch pair)"; $B --batches-per-launch 32 2>/dev/null | show
echo "== K=32 2 partitions"; $B --batches-per-launch 32 --partitions 2 2>/dev/null | show
echo "== K=32 plain"; $B --batches-per-launch 32 --partitions 0 2>/dev/null | show
echo "== K=24"; $B --batches-per-launch 24 2>/dev/null | show
echo "== K=64 (two launch pairs per call)"; $B --batches-per-launch 64 2>/dev/null | show
echo "== K=16 again"; $B 2>/dev/null | show
