#!/bin/bash
# After the ring keeps four lanes per signature: the streamed-rounds bisect, the GPU suite, the driver's bench command.
mkdir -p gpurun_out/r3g
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g
timeout 300 python tools/streamed_probe.py bisect > $O/streamed_bisect.txt 2> $O/streamed_bisect.err; echo "bisect rc=$?"; cat $O/streamed_bisect.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r3_bench_default.json.txt 2> $O/r3_bench_default.err; echo "bench rc=$?"
python - <<'P'
import json
for l in open('gpurun_out/r3g/r3_bench_default.json.txt'):
    try: j = json.loads(l)
    except Exception: continue
    print('headline %.2f M' % (j['value'] / 1e6), j['roofline']['kernel'], j['roofline']['avg_kernel_ms'], 'traffic', j['roofline'].get('traffic'))
    q = j.get('qc_verify_latency', {}); print('qc', q.get('p50_ms_device_resident'), q.get('p50_ms_host_api'), q.get('committee_cached'))
    print('streamed', {k: (round(v.get('signatures_per_s', 0) / 1e6, 1), v.get('submit_to_verdict_p50_ms')) for k, v in j.get('streamed_rounds', {}).items() if isinstance(v, dict)})
    for r in j.get('end_to_end', {}).get('by_threads', []): print('e2e', r['host_threads'], r['in_flight_per_thread'], r['batches_per_launch'], '%.1f M' % (r['value'] / 1e6), r['call_p50_ms'])
P
