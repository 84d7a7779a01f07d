#!/bin/bash
# The sliver of large one-lane calls and the concurrency hint: parity at the sizes concerned, the hint A/B, the sliver-size sweep
# (profiles/r3_concurrent_ab.txt sections 2 and 3).   gpurun --timeout 900 -- 'bash tools/gpu_r3_conc.sh'
mkdir -p gpurun_out/r3j
export HS_SYNTH_CACHE=/tmp/hs_synth
timeout 600 python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_rlc.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/concurrent_ab.py 6667,66667 > gpurun_out/r3j/concurrent_ab2.md 2> gpurun_out/r3j/concurrent_ab2.err; echo "ab rc=$?"; cat gpurun_out/r3j/concurrent_ab2.md
for tm in 0 65536 0 65536; do
  echo "-- HS_ED25519_TAIL_MAX=$tm"
  HS_SWEEP_NO_HINT=1 HS_ED25519_TAIL_MAX=$tm HS_SWEEP_SIZES=65536,66667,67584,69632,73728,81920,98304 timeout 300 python tools/r2_sweeps.py 2>/dev/null | grep -E "^\| [0-9]"
done | tee gpurun_out/r3j/tail_sweep.md
