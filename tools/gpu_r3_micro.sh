#!/bin/bash
# same-box A/B of the last field-arithmetic touches (to_p3 operand order, 38 f as a shift): previous build against this one,
# headline only, alternating; then the GPU suite on the new build
mkdir -p gpurun_out/r3p
export HS_SYNTH_CACHE=/tmp/hs_synth
for i in 1 2 3; do
  for lib in build/prev/libhs_ed25519.so hotstuff_amd/libhs_ed25519.so; do
    HS_ED25519_LIB=$GRAFT_REPO_ROOT/$lib timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu --no-extras 2>/dev/null | python -c "
import json,sys
j=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
print('$lib', '%.2f M' % (j['value']/1e6), j['roofline']['avg_kernel_ms'], j['kernels_ms']['k_prepare'])"
  done
done | tee gpurun_out/r3p/micro_ab.txt
timeout 600 python -m pytest tests -q -x -m gpu > gpurun_out/r3p/pytest_gpu.txt 2>&1; grep -E "passed|failed" gpurun_out/r3p/pytest_gpu.txt | tail -1
