#!/bin/bash
mkdir -p gpurun_out/r3
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3
echo "== full GPU suite"; timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu_g.txt 2>&1; echo "rc=$?"; grep -E " passed| failed" $O/pytest_gpu_g.txt | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_gpu_g.txt | head -20
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench default"; timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default_g.json.txt 2> $O/bench_default_g.err; echo "rc=$?"
python - <<'P'
import json
for l in open('gpurun_out/r3/bench_default_g.json.txt'):
    try: j = json.loads(l)
    except Exception: continue
    print('headline %.2f M int_roofline %s kernels %s' % (j['value'] / 1e6, (j.get('int_roofline') or {}).get('frac'), j['kernels_ms']))
    e = j.get('end_to_end', {})
    for r in e.get('by_threads', []): print('  e2e', r['host_threads'], r['in_flight_per_thread'], r['batches_per_launch'], '%.1f M' % (r['value'] / 1e6), r['call_p50_ms'])
    for r in e.get('caller_pinned', {}).get('rows', []): print('  pinned', r['host_threads'], r['in_flight_per_thread'], r['batches_per_launch'], '%.1f M' % (r['value'] / 1e6), r['call_p50_ms'])
    print('  multi_process', [(r.get('processes'), r.get('tenancy_hint'), r.get('value'), r.get('vs_one_process')) for r in j.get('multi_process', {}).get('rows', [])])
    q = j.get('qc_verify_latency', {}); print('  qc', q.get('p50_ms_device_resident'), q.get('p50_ms_host_api'), q.get('committee_cached'))
    print('  cached', j['committee_cached']['value'], j['committee_cached']['large_committee'].get('avg_kernel_ms'))
P
