#!/bin/bash
mkdir -p gpurun_out/r3
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3
echo "== full GPU suite"; timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu_h.txt 2>&1; echo "rc=$?"; grep -E " passed| failed" $O/pytest_gpu_h.txt | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_gpu_h.txt | head -20
echo "== headline by kernel form and batches per launch"
for cfg in "auto 16" "lane 4" "lane 8" "pairs 4" "pairs 8" "fused 2" "fused 4" "fused 1" "lane 2"; do
  set -- $cfg
  timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu --no-extras --kernel $1 --batches-per-launch $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('kernel $1 K=$2: %.2f M  %s %.4f ms prepare %s partitions %s' % (j['value']/1e6, j['roofline']['kernel'], j['roofline']['avg_kernel_ms'], j['kernels_ms']['k_prepare'], j['config']['cu_partitions']))"
done
echo "== sweeps (sections 1, 2)"
HS_SWEEP_SKIP3=1 timeout 900 python tools/r2_sweeps.py > $O/r3_sweeps_h.md 2> $O/r3_sweeps_h.err; echo "rc=$?"; cat $O/r3_sweeps_h.md
