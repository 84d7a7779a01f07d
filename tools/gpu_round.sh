#!/bin/bash
# One gpurun visit: the GPU suite, smoke(), and the bench line of the driver's command (summary on stdout, the JSON line in
# gpurun_out/bench_default.json.txt).   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh'
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_all.txt 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/pytest_gpu_all.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/bench_default.err | tee gpurun_out/bench_default.json.txt | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    q = j.get('qc_verify_latency', {}); e = j.get('end_to_end', {})
    print('headline %.2f M  %s %.4f ms (prepare %s)  int_roofline %.3f  K %d on %s partitions | cached %.1f M  66,667-signature launch %.4f ms | qc cold %.4f / %.4f ms, cached %.4f / %.4f ms (device / host API) | e2e %.1f M | cpu %.2f M' % (
        j['value'] / 1e6, j['roofline']['kernel'], j['roofline']['avg_kernel_ms'], j['kernels_ms']['k_prepare'], (j['int_roofline'] or {}).get('frac', 0),
        j['config']['batches_per_launch'], j['config']['cu_partitions'], j['committee_cached']['value'] / 1e6, j['committee_cached']['large_committee']['avg_kernel_ms'],
        q.get('p50_ms_device_resident', 0), q.get('p50_ms_host_api', 0), q.get('committee_cached', {}).get('p50_ms_device_resident', 0),
        q.get('committee_cached', {}).get('p50_ms_host_api', 0), e.get('value', 0) / 1e6, j.get('cpu_baseline', {}).get('value', 0) / 1e6))
"
