#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rlc.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | head -5
show() { python - "$1" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select name, total_calls, average from top_kernels"):
    if "k_rlc" in r[0]:
        print("   %-22s calls %4d avg %8.1f us" % (r[0].split("(")[0], r[1], r[2]))
PY
}
for n in 6667 66667; do
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/p_$n -o rlc -- python tools/rlc_probe.py rlc $n > gpurun_out/p$n.log 2>&1; grep "^mode" gpurun_out/p$n.log
  show gpurun_out/p_$n/rlc_results.db; rm -rf gpurun_out/p_$n
done
timeout 300 python tools/rlc_probe.py rlc 66667 cached | grep "^mode"
