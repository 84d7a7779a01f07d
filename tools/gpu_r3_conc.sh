#!/bin/bash
mkdir -p gpurun_out/r3j
export HS_SYNTH_CACHE=/tmp/hs_synth
timeout 600 python -m pytest "tests/test_gpu_baseline_sizes.py::test_cold_and_cached_vs_oracle_at_baseline_size[66667]" -x -q -m gpu 2>&1 | tail -40
for side in 0 1 0 1; do
  echo "-- HS_ED25519_TAIL_SIDE=$side"
  HS_SWEEP_NO_HINT=1 HS_ED25519_TAIL_SIDE=$side HS_SWEEP_SIZES=66667,67584 timeout 300 python tools/r2_sweeps.py 2>/dev/null | grep -E "^\| [0-9]"
done | tee gpurun_out/r3j/tail_form_ab.md
