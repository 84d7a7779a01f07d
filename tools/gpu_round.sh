#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json.txt 2>gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.err; wc -c gpurun_out/bench_default.json.txt
