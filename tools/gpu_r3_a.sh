#!/bin/bash
# Round-3 GPU visit A: the GPU suite, smoke, the driver's bench command, the `--gpus 2` rehearsal on one GPU, the
# counter calibration and the solo-launch counter passes.   gpurun --timeout 1500 -- 'bash tools/gpu_r3_a.sh'
mkdir -p gpurun_out/r3
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -5 $O/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench default"; timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json.txt 2> $O/bench_default.err; echo "rc=$?"
python - <<'P'
import json
for l in open('gpurun_out/r3/bench_default.json.txt'):
    try: j = json.loads(l)
    except Exception: continue
    print('headline %.2f M int_roofline %s' % (j['value'] / 1e6, (j.get('int_roofline') or {}).get('frac')))
    e = j.get('end_to_end', {})
    for r in e.get('by_threads', []): print('  e2e', r['host_threads'], r['in_flight_per_thread'], r['batches_per_launch'], '%.1f M' % (r['value'] / 1e6), r['call_p50_ms'])
    print('  pinned', json.dumps(e.get('caller_pinned'))[:600])
    print('  multi_process', json.dumps(j.get('multi_process'))[:1500])
    q = j.get('qc_verify_latency', {}); print('  qc', q.get('p50_ms_device_resident'), q.get('p50_ms_host_api'), q.get('committee_cached'))
P
echo "== rehearsal: python bench.py --gpus 2 on ONE GPU (ranks share the device, gloo control plane)"
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --large-committee 20000 > $O/bench_gpus2_rehearsal.json.txt 2> $O/bench_gpus2_rehearsal.err; echo "rc=$?"
head -c 1500 $O/bench_gpus2_rehearsal.json.txt; echo; grep -E "^bench:" $O/bench_gpus2_rehearsal.err | head
python -c "
import json
j=json.loads(open('$O/bench_gpus2_rehearsal.json.txt').read().strip().splitlines()[-1])
print('n_gpus', j['n_gpus'], 'value %.1f M' % (j['value']/1e6), j['config'].get('ranks'))
print('sharded_committee', json.dumps(j.get('sharded_committee'))[:800])
print('sharded_c_abi', json.dumps(j.get('sharded_c_abi'))[:1800])
"
echo "== rehearsal: 1 rank, forced exchange (RCCL, one rank) + sharded C ABI"
HS_BENCH_FORCE_EXCHANGE=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu --no-qc --no-cached --large-committee 20000 > $O/bench_force_exchange.json.txt 2> $O/bench_force_exchange.err; echo "rc=$?"
python -c "
import json
j=json.loads(open('$O/bench_force_exchange.json.txt').read().strip().splitlines()[-1])
print(j['config'].get('ranks')); print('sharded_committee', json.dumps(j.get('sharded_committee'))[:600]); print('sharded_c_abi', json.dumps(j.get('sharded_c_abi'))[:1200])
"
echo "== counter calibration"
timeout 300 tools/ubench_gather > $O/ubench_known.json; echo "rc=$?"; cat $O/ubench_known.json | head -c 600; echo
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d $O/cal_$c -o cal -- tools/ubench_gather > /dev/null 2> $O/cal_$c.err; echo "cal $c rc=$?"
done
python tools/pmc_calibrate.py $O/ubench_known.json $O/cal_FETCH_SIZE/cal_results.db $O/cal_WRITE_SIZE/cal_results.db > $O/r3_pmc_calibration.json; python -c "
import json; print(json.dumps(json.load(open('$O/r3_pmc_calibration.json'))['factors'], indent=1))"
echo "== solo-launch counters"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $O/solo_$c -o p -- python tools/pmc_solo.py > $O/solo_$c.out 2> $O/solo_$c.err; echo "solo $c rc=$?"
  python tools/rocpd_pmc.py $O/solo_$c/p_results.db --by-grid k_ > $O/r3_pmc_solo_$c.json
done
python - <<'P'
import json
f = json.load(open('gpurun_out/r3/r3_pmc_solo_FETCH_SIZE.json')); w = json.load(open('gpurun_out/r3/r3_pmc_solo_WRITE_SIZE.json'))
for k in sorted(f):
    print(k, 'FETCH KB %.1f' % f[k].get('FETCH_SIZE', {}).get('avg_per_dispatch', 0), 'x', f[k].get('FETCH_SIZE', {}).get('dispatches'),
          ' WRITE KB %.1f' % w.get(k, {}).get('WRITE_SIZE', {}).get('avg_per_dispatch', 0))
P
rm -rf $O/cal_FETCH_SIZE $O/cal_WRITE_SIZE $O/solo_FETCH_SIZE $O/solo_WRITE_SIZE
