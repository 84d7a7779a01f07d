#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== full suite"; timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/t_all.log 2>&1; echo "rc=$?"; grep -E "passed|failed" gpurun_out/t_all.log; grep -E "^FAILED" gpurun_out/t_all.log | head
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
print(d['value'], d['roofline']['avg_kernel_ms'])
c=d['committee_cached']; print({k:c[k] for k in ('value','avg_kernel_ms','keyset_build_ms','keyset_hbm_bytes')}); print(json.dumps(c['large_committee'])[:600])
q=d['qc_verify_latency']; print(q['p50_ms_host_api'], q['p50_ms_device_resident'], q['committee_cached'])
s=d['streamed_rounds']; print({k:(v['signatures_per_s'],v['submit_to_verdict_p50_ms']) for k,v in s.items() if isinstance(v,dict)})
print(d['small_batch_crossover']['rows'][:3])
PY
