#!/bin/bash
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tools/half_probe | grep -E "scalar stage|per "
echo "== latency"; HS_SWEEP_SIZES=2048,6667,16384 timeout 300 python tools/r2_sweeps.py 2>/dev/null | grep -E "^\| [0-9]"
for cfg in "auto 16" "pairs 4" "fused 2"; do
  set -- $cfg
  timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu --no-extras --kernel $1 --batches-per-launch $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('kernel $1 K=$2: %.2f M  %s %.4f ms prepare %s' % (j['value']/1e6, j['roofline']['kernel'], j['roofline']['avg_kernel_ms'], j['kernels_ms']['k_prepare']))"
done
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py tests/test_gpu_fuzz_apis.py -q 2>&1 | grep -E " passed| failed"
