#!/bin/bash
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in hotstuff_amd/libhs_ed25519.so build/libhs_nopf.so hotstuff_amd/libhs_ed25519.so build/libhs_nopf.so; do
  HS_ED25519_LIB=$GRAFT_REPO_ROOT/$lib timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('$lib: %.2f M  %s %.4f ms prepare %s' % (j['value']/1e6, j['roofline']['kernel'], j['roofline']['avg_kernel_ms'], j['kernels_ms']['k_prepare']))"
done
for lib in hotstuff_amd/libhs_ed25519.so build/libhs_nopf.so; do
  echo "-- $lib"; HS_ED25519_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python tools/lane_probe.py 131072 262144 2>/dev/null
done
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py -q 2>&1 | grep -E " passed| failed"
