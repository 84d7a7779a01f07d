#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_tickets.py tests/test_gpu_parity.py tests/test_gpu_cpp_mirror.py tests/test_gpu_ring.py tests/test_gpu_ingest.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
show() { python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    q = j.get('qc_verify_latency', {})
    e = j.get('end_to_end', {})
    print('headline %.2f M  kernel %s %.4f ms (prepare %s) int_roofline %.3f  K %d part %s | cached %.1f M  large %.4f ms | qc cold dev %.4f host %.4f cached dev %.4f | e2e %.1f M (%s thr x %s tickets x %s per launch, p50 %s ms)' % (
        j['value'] / 1e6, j['roofline']['kernel'], j['roofline']['avg_kernel_ms'], j['kernels_ms']['k_prepare'], (j['int_roofline'] or {}).get('frac', 0), j['config']['batches_per_launch'], j['config']['cu_partitions'],
        j['committee_cached']['value'] / 1e6, j['committee_cached']['large_committee']['avg_kernel_ms'], q.get('p50_ms_device_resident', 0), q.get('p50_ms_host_api', 0),
        q.get('committee_cached', {}).get('p50_ms_device_resident', 0), e.get('value', 0) / 1e6, e.get('host_threads'), e.get('in_flight_per_thread'), e.get('batches_per_launch'), e.get('call_p50_ms')))
    for r in e.get('by_threads', []): print('   e2e', r['host_threads'], r['in_flight_per_thread'], r['batches_per_launch'], round(r['value'] / 1e6, 1), r['call_p50_ms'])
"; }
B="timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu"
echo "== default"; $B 2>gpurun_out/bench_default.err | tee gpurun_out/bench_default.json.txt | show
echo "== pairs K=4"; $B --kernel pairs --batches-per-launch 4 --no-qc --no-cached 2>/dev/null | show
