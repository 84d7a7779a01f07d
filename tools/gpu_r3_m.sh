#!/bin/bash
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== latency, eight lanes per signature on (default) / off"
for v in 8192 0 8192 0; do
  echo "-- HS_ED25519_OCT_MAX_BATCH=$v"; HS_ED25519_OCT_MAX_BATCH=$v HS_SWEEP_SIZES=1024,4096,6667,8192 timeout 300 python tools/r2_sweeps.py 2>/dev/null | grep -E "^\| [0-9]"
done
echo "== parity"; timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E " passed| failed|^FAILED"
