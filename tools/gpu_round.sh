#!/bin/bash
mkdir -p gpurun_out/r2
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d gpurun_out/r2/pmcl_$c -o p -- python bench.py --steps 2 --warmup 1 --no-cpu --no-qc > /dev/null 2> gpurun_out/r2/pmcl_$c.err; echo rc=$?
  python tools/rocpd_pmc.py gpurun_out/r2/pmcl_$c/p_results.db --by-grid k_verify_cached > gpurun_out/r2/r2_pmc_${c}_cached.json; cat gpurun_out/r2/r2_pmc_${c}_cached.json | tr -d '\n '; echo
  rm -rf gpurun_out/r2/pmcl_$c
done
