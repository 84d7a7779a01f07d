#!/bin/bash
mkdir -p gpurun_out/r3
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3
echo "== half_probe"; timeout 120 tools/half_probe | head -12
echo "== latency"; HS_SWEEP_SIZES=1024,6667,16384,66667,69632 timeout 300 python tools/r2_sweeps.py 2>/dev/null | grep -E "^\| [0-9]"
echo "== headline"; timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('%.2f M  %s %.4f ms prepare %s' % (j['value']/1e6, j['roofline']['kernel'], j['roofline']['avg_kernel_ms'], j['kernels_ms']['k_prepare']))"
echo "== SQ counters (headline)"; timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq -o p -- python bench.py --steps 3 --warmup 1 --no-extras --no-cpu > /dev/null 2> $O/pmc_sq.err; echo rc=$?
python tools/rocpd_pmc.py $O/pmc_sq/p_results.db k_verify k_prepare > $O/r3_pmc_sq_headline.json; cat $O/r3_pmc_sq_headline.json | tr -d '\n '; echo
echo "== SQ counters (fused, lone wavefronts: QC 6,667)"; timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq2 -o p -- python tools/rlc_probe.py qc > /dev/null 2> $O/pmc_sq2.err; echo rc=$?
python tools/rocpd_pmc.py $O/pmc_sq2/p_results.db k_verify > $O/r3_pmc_sq_qc.json; cat $O/r3_pmc_sq_qc.json | tr -d '\n '; echo
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $O/solo_$c -o p -- python tools/pmc_solo.py batches lane > $O/solo_$c.out 2> $O/solo_$c.err; echo "solo $c rc=$?"
  python tools/rocpd_pmc.py $O/solo_$c/p_results.db --by-grid k_verify k_prepare > $O/r3_pmc_solo2_$c.json
done
python - <<'P'
import json
f = json.load(open('gpurun_out/r3/r3_pmc_solo2_FETCH_SIZE.json')); w = json.load(open('gpurun_out/r3/r3_pmc_solo2_WRITE_SIZE.json'))
for k in sorted(f):
    print(k, 'FETCH KB %.1f' % f[k].get('FETCH_SIZE', {}).get('avg_per_dispatch', 0), 'x', f[k].get('FETCH_SIZE', {}).get('dispatches'),
          ' WRITE KB %.1f' % w.get(k, {}).get('WRITE_SIZE', {}).get('avg_per_dispatch', 0))
P
rm -rf $O/pmc_sq $O/pmc_sq2 $O/solo_FETCH_SIZE $O/solo_WRITE_SIZE
echo "== rehearsal test"; timeout 900 python -m pytest tests/test_gpu_bench_ranks.py -q 2>&1 | tail -3
