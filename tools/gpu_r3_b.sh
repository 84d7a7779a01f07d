#!/bin/bash
# Round-3 GPU visit B: parity suite on the half-size cold path, the driver's bench command, kernel-form sweeps.
mkdir -p gpurun_out/r3
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_b.txt 2>&1; echo "rc=$?"; grep -E "passed|failed|Error|error" $O/pytest_gpu_b.txt | tail -5
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench default"; timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default_b.json.txt 2> $O/bench_default_b.err; echo "rc=$?"
python - <<'P'
import json
for l in open('gpurun_out/r3/bench_default_b.json.txt'):
    try: j = json.loads(l)
    except Exception: continue
    print('headline %.2f M int_roofline %s kernels %s' % (j['value'] / 1e6, (j.get('int_roofline') or {}).get('frac'), j['kernels_ms']))
    e = j.get('end_to_end', {})
    for r in e.get('by_threads', []): print('  e2e', r['host_threads'], r['in_flight_per_thread'], r['batches_per_launch'], '%.1f M' % (r['value'] / 1e6), r['call_p50_ms'])
    for r in e.get('caller_pinned', {}).get('rows', []): print('  pinned', r['host_threads'], r['in_flight_per_thread'], r['batches_per_launch'], '%.1f M' % (r['value'] / 1e6), r['call_p50_ms'])
    print('  multi_process', [(r.get('processes'), r.get('tenancy_hint'), r.get('value'), r.get('vs_one_process')) for r in j.get('multi_process', {}).get('rows', [])])
    q = j.get('qc_verify_latency', {}); print('  qc', q.get('p50_ms_device_resident'), q.get('p50_ms_host_api'), q.get('committee_cached'))
    print('  cached', j['committee_cached']['value'], j['committee_cached']['large_committee'].get('avg_kernel_ms'))
    print('  rlc_ab', json.dumps(j.get('rlc_ab')))
    print('  streamed', {k: (v.get('signatures_per_s'), v.get('submit_to_verdict_p50_ms')) for k, v in j.get('streamed_rounds', {}).items() if isinstance(v, dict)})
    print('  crossover', j.get('small_batch_crossover', {}).get('rows', [])[:2])
P
echo "== headline by kernel form and batches per launch"
for cfg in "auto 16" "lane 8" "lane 16 0" "pairs 4" "pairs 8" "fused 2" "fused 4" "fused 1"; do
  set -- $cfg
  extra=""; [ -n "$3" ] && extra="--partitions $3"
  timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu --no-extras --kernel $1 --batches-per-launch $2 $extra 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('kernel $1 K=$2 $3: %.2f M  %s %.4f ms prepare %s partitions %s' % (j['value']/1e6, j['roofline']['kernel'], j['roofline']['avg_kernel_ms'], j['kernels_ms']['k_prepare'], j['config']['cu_partitions']))"
done
echo "== sweeps (sections 1, 2)"
HS_SWEEP_SKIP3=1 timeout 900 python tools/r2_sweeps.py > $O/r3_sweeps_b.md 2> $O/r3_sweeps_b.err; echo "rc=$?"; cat $O/r3_sweeps_b.md
