#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_keyset.py -x -q -m gpu -k "key_cache_batches or several or keyset" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
for rep in 1 2; do
echo "== rlc sweep default (launch_bounds 64,3)"; HS_SWEEP_SIZES=24576,49152,66667,131072 timeout 300 python tools/r2_sweeps.py 2>/dev/null | grep "^| [0-9]"
echo "== rlc sweep variant (launch_bounds 64,2)"; HS_ED25519_LIB=build/variant/libhs_ed25519.so HS_SWEEP_SIZES=24576,49152,66667,131072 timeout 300 python tools/r2_sweeps.py 2>/dev/null | grep "^| [0-9]"
done
showc() { python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('headline %.1f M | cached %.1f M  kernel avg %.4f ms  K %d' % (j['value'] / 1e6, j['committee_cached']['value'] / 1e6, j['committee_cached']['avg_kernel_ms'], j['committee_cached']['batches_per_launch']))
"; }
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-qc"
echo "== cached lane2r K=4"; $B 2>/dev/null | showc
echo "== cached lane K=16"; $B --kernel lane --cached-batches-per-launch 16 2>/dev/null | showc
echo "== cached lane K=8"; $B --kernel lane --cached-batches-per-launch 8 2>/dev/null | showc
echo "== cached lane2r K=16"; $B --cached-batches-per-launch 16 2>/dev/null | showc
