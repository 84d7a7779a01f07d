#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_keyset.py tests/test_gpu_fuzz_apis.py tests/test_gpu_rlc.py tests/test_gpu_baseline_sizes.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
show() { python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    q = j.get('qc_verify_latency', {})
    print('headline %.2f M  kernel %s %.4f ms  int_roofline %.3f  K %d | cached %.1f M  large %.4f ms | qc cold dev %.4f host %.4f cached dev %.4f | e2e %.1f M' % (
        j['value'] / 1e6, j['roofline']['kernel'], j['roofline']['avg_kernel_ms'], (j['int_roofline'] or {}).get('frac', 0), j['config']['batches_per_launch'],
        j['committee_cached']['value'] / 1e6, j['committee_cached']['large_committee']['avg_kernel_ms'], q.get('p50_ms_device_resident', 0), q.get('p50_ms_host_api', 0),
        q.get('committee_cached', {}).get('p50_ms_device_resident', 0), j['end_to_end']['value'] / 1e6))
"; }
B="timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu"
echo "== fused K=2, cached K=2"; $B --kernel fused --batches-per-launch 2 2>/dev/null | show
echo "== pairs K=4, cached K=4"; $B --kernel pairs --batches-per-launch 4 --cached-batches-per-launch 4 2>/dev/null | show
echo "== pairs K=4, cached K=8"; $B --kernel pairs --batches-per-launch 4 --cached-batches-per-launch 8 2>/dev/null | show
