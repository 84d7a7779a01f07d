#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_keyset.py tests/test_gpu_fuzz_apis.py tests/test_gpu_baseline_sizes.py tests/test_gpu_tickets.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
show() { python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    q = j.get('qc_verify_latency', {})
    e = j.get('end_to_end', {})
    print('headline %.2f M  kernel %s %.4f ms  int_roofline %.3f  K %d | cached %.1f M  large %.4f ms | qc cold dev %.4f host %.4f cached dev %.4f | e2e %.1f M (%s thr x %s tickets x %s per launch, p50 %s ms)' % (
        j['value'] / 1e6, j['roofline']['kernel'], j['roofline']['avg_kernel_ms'], (j['int_roofline'] or {}).get('frac', 0), j['config']['batches_per_launch'],
        j['committee_cached']['value'] / 1e6, j['committee_cached']['large_committee']['avg_kernel_ms'], q.get('p50_ms_device_resident', 0), q.get('p50_ms_host_api', 0),
        q.get('committee_cached', {}).get('p50_ms_device_resident', 0), e.get('value', 0) / 1e6, e.get('host_threads'), e.get('in_flight_per_thread'), e.get('batches_per_launch'), e.get('call_p50_ms')))
"; }
B="timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu"
for rep in 1 2; do
echo "== default (q_sel = v_bfi)"; $B 2>/dev/null | show
echo "== variant (q_sel = v_cndmask)"; HS_ED25519_LIB=build/variant/libhs_ed25519.so $B 2>/dev/null | show
done
echo "== fused K=2 default lib"; $B --kernel fused --batches-per-launch 2 --no-qc --no-cached 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('headline %.2f M kernel %.4f ms' % (j['value'] / 1e6, j['roofline']['avg_kernel_ms']))"
