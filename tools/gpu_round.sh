#!/bin/bash
export HS_SYNTH_CACHE=/tmp/hs_synth
export HS_SWEEP_SIZES=8192,16384,24576,32768,66667,131072
echo "== old (before pipelining / split / wave atomics)"; HS_ED25519_LIB=build/variant_old/libhs_ed25519.so timeout 600 python tools/r2_sweeps.py 2>/dev/null | grep "^| [0-9]"
echo "== new"; timeout 600 python tools/r2_sweeps.py 2>/dev/null | grep "^| [0-9]"
echo "== old again"; HS_ED25519_LIB=build/variant_old/libhs_ed25519.so timeout 600 python tools/r2_sweeps.py 2>/dev/null | grep "^| [0-9]"
