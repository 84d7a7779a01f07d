#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HS_SWEEP_SIZES=16384,32768,66667,131072
echo "== default (accum 2 waves/SIMD)"; timeout 600 python tools/r2_sweeps.py 2>/dev/null | grep "^| [0-9]"
echo "== variant (accum 3 waves/SIMD)"; HS_ED25519_LIB=build/variant/libhs_ed25519.so timeout 600 python tools/r2_sweeps.py 2>/dev/null | grep "^| [0-9]"
echo "== default again"; timeout 600 python tools/r2_sweeps.py 2>/dev/null | grep "^| [0-9]"
unset HS_SWEEP_SIZES
echo "== sustained rounds, 10 s each"; timeout 300 python tools/ring_sweep.py --help 2>&1 | head -5
