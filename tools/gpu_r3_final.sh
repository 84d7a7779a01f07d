#!/bin/bash
# Round-3 evidence visit: same-box A/B against the round-2 tree (build/r2tree), the driver's bench command, rocprofv3 kernel
# stats of the headline, SQ counters, solo HBM counters (calibrated), the 2-rank rehearsal, sweeps.
#   gpurun --timeout 2400 -- 'bash tools/gpu_r3_final.sh'
O=${HS_FINAL_OUT:-gpurun_out/r3f}
mkdir -p $O
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
summ() { python - "$1" <<'P'
import json, sys
for l in open(sys.argv[1]):
    try: j = json.loads(l)
    except Exception: continue
    q = j.get('qc_verify_latency', {}); e = j.get('end_to_end', {})
    print('  headline %.2f M  %s %.4f ms (prepare %s) int_roofline %s' % (j['value'] / 1e6, j['roofline']['kernel'], j['roofline']['avg_kernel_ms'], j['kernels_ms']['k_prepare'], (j.get('int_roofline') or {}).get('frac')))
    print('  traffic', j['roofline'].get('traffic'), (j['roofline'].get('traffic_source') or '')[:90])
    for r in e.get('by_threads', []): print('  e2e', r['host_threads'], r['in_flight_per_thread'], r['batches_per_launch'], '%.1f M' % (r['value'] / 1e6), r['call_p50_ms'])
    for r in e.get('caller_pinned', {}).get('rows', []): print('  pinned', r['host_threads'], r['in_flight_per_thread'], r['batches_per_launch'], '%.1f M' % (r['value'] / 1e6), r['call_p50_ms'])
    print('  qc cold %s / %s ms, cached %s (device / host API)' % (q.get('p50_ms_device_resident'), q.get('p50_ms_host_api'), q.get('committee_cached')))
    c = j.get('committee_cached', {}); print('  cached %.1f M; 66,667 launch %s ms' % (c.get('value', 0) / 1e6, c.get('large_committee', {}).get('avg_kernel_ms')))
    print('  rlc_ab', json.dumps(j.get('rlc_ab')))
    print('  streamed', {k: (round(v.get('signatures_per_s', 0) / 1e6, 1), v.get('submit_to_verdict_p50_ms')) for k, v in j.get('streamed_rounds', {}).items() if isinstance(v, dict)})
    print('  multi_process', [(r.get('processes'), r.get('tenancy_hint'), r.get('value'), r.get('vs_one_process')) for r in j.get('multi_process', {}).get('rows', [])])
    print('  crossover', j.get('small_batch_crossover', {}).get('measured_crossover_n'), j.get('small_batch_crossover', {}).get('rows', [{}])[0])
P
}
echo "== round-2 tree, same box"; (cd build/r2tree && timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu > $GRAFT_REPO_ROOT/$O/r2tree_bench.json.txt 2> $GRAFT_REPO_ROOT/$O/r2tree_bench.err); echo "rc=$?"; summ $O/r2tree_bench.json.txt
echo "== round 3, the driver's command"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/r3_bench_default.json.txt 2> $O/r3_bench_default.err; echo "rc=$?"; summ $O/r3_bench_default.json.txt
echo "== kernel trace, headline only"; timeout 600 rocprofv3 --kernel-trace --stats -d $O/kth -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu --no-extras > $O/r3_bench_headline_under_rocprof.json.txt 2> $O/kth.err; echo rc=$?
python tools/rocpd_summary.py $O/kth/kt_results.db > $O/r3_bench_headline_kernel_stats.csv; head -4 $O/r3_bench_headline_kernel_stats.csv | cut -c1-120
python -c "
import json
for l in open('$O/r3_bench_headline_under_rocprof.json.txt'):
    try: j = json.loads(l)
    except Exception: continue
    print('under rocprof: value %.2f M, events avg_kernel_ms %.4f (prepare %s) over %d launches' % (j['value']/1e6, j['roofline']['avg_kernel_ms'], j['kernels_ms']['k_prepare'], j['roofline']['launches_timed']))
"
echo "== SQ counters (headline)"; timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq -o p -- python bench.py --steps 3 --warmup 1 --no-extras --no-cpu > /dev/null 2> $O/pmc_sq.err; echo rc=$?
python tools/rocpd_pmc.py $O/pmc_sq/p_results.db k_verify k_prepare > $O/r3_pmc_sq_headline.json; cat $O/r3_pmc_sq_headline.json | tr -d '\n '; echo
echo "== SQ counters (QC 6,667 cold + cached, lone wavefronts)"; timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq2 -o p -- python tools/rlc_probe.py qc > /dev/null 2> $O/pmc_sq2.err; echo rc=$?
python tools/rocpd_pmc.py $O/pmc_sq2/p_results.db k_verify > $O/r3_pmc_sq_qc.json; cat $O/r3_pmc_sq_qc.json | tr -d '\n '; echo
echo "== solo HBM counters"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $O/solo_$c -o p -- python tools/pmc_solo.py > $O/solo_$c.out 2> $O/solo_$c.err; echo "solo $c rc=$?"
  python tools/rocpd_pmc.py $O/solo_$c/p_results.db --by-grid k_ > $O/r3_pmc_solo_$c.json
done
python tools/pmc_traffic_report.py $O/r3_pmc_solo_FETCH_SIZE.json $O/r3_pmc_solo_WRITE_SIZE.json profiles/r3_pmc_calibration.json > $O/r3_pmc_traffic.json
python -c "
import json
j=json.load(open('$O/r3_pmc_traffic.json'))
for k,v in j['kernels'].items(): print(' ', k, 'read MB %.1f write MB %.1f per sig %s x algorithmic %s' % (v['hbm_read_bytes_per_launch']/1e6, v['hbm_write_bytes_per_launch']/1e6, v['hbm_bytes_per_signature'], v.get('traffic_over_algorithmic')))
"
rm -rf $O/kth $O/pmc_sq $O/pmc_sq2 $O/solo_FETCH_SIZE $O/solo_WRITE_SIZE
echo "== rehearsal: python bench.py --gpus 2 on ONE GPU"; timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --large-committee 20000 > $O/r3_bench_gpus2_rehearsal.json.txt 2> $O/r3_bench_gpus2_rehearsal.err; echo "rc=$?"
python -c "
import json
j=json.loads(open('$O/r3_bench_gpus2_rehearsal.json.txt').read().strip().splitlines()[-1])
print('n_gpus', j['n_gpus'], 'value %.1f M' % (j['value']/1e6), j['config'].get('ranks'))
print('sharded_committee', json.dumps(j.get('sharded_committee'))[:700])
print('sharded_c_abi', json.dumps(j.get('sharded_c_abi'))[:1600])
"
echo "== rehearsal: 1 rank, forced exchange (RCCL one rank)"; HS_BENCH_FORCE_EXCHANGE=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu --no-qc --no-cached --large-committee 20000 > $O/r3_bench_force_exchange.json.txt 2> $O/r3_bench_force_exchange.err; echo "rc=$?"
if [[ "$HS_FINAL_SKIP" != *sweeps* ]]; then echo "== sweeps"; timeout 1200 python tools/r2_sweeps.py > $O/r3_sweeps.md 2> $O/r3_sweeps.err; echo "rc=$?"; cat $O/r3_sweeps.md; fi
echo "== tail rows"; HS_SWEEP_SIZES=65536,66667,67584,69632,81920,98304 timeout 300 python tools/r2_sweeps.py 2>/dev/null | grep -E "^\| [0-9]" | tee $O/r3_tail_rows.md
echo "== streamed 10 s"; timeout 120 python tools/streamed_10s.py > $O/r3_streamed_10s.md 2>/dev/null; tail -6 $O/r3_streamed_10s.md
if [[ "$HS_FINAL_SKIP" != *suite* ]]; then echo "== full GPU suite"; timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "rc=$?"; grep -E " passed| failed" $O/pytest_gpu.txt | tail -2; fi
