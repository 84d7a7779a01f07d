#!/bin/bash
# One of the round's eight two-rank rehearsals hit the 150 s limit of the one-process extra (sharded_c_abi) without a diagnosis:
# repeat the rehearsal until it recurs (the tool now prints stage markers and its threads' stacks before the limit).
#   gpurun --timeout 900 -- 'bash tools/gpu_r3_rehearse_loop.sh 20'
mkdir -p gpurun_out/r3l
export HS_SYNTH_CACHE=/tmp/hs_synth
N=${1:-15}
for i in $(seq 1 $N); do
  t0=$SECONDS
  timeout 400 python bench.py --gpus 2 --steps 3 --warmup 1 --large-committee 20000 > gpurun_out/r3l/reh.json.txt 2> gpurun_out/r3l/reh.err
  rc=$?
  python - gpurun_out/r3l/reh.json.txt $i $rc $((SECONDS - t0)) <<'P'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("rehearsal %s rc=%s wall %s s: no line (%s)" % (sys.argv[2], sys.argv[3], sys.argv[4], e)); sys.exit(0)
bad = {k: v for k, v in j.items() if isinstance(v, dict) and "error" in v}
print("rehearsal %s rc=%s wall %s s: value %.1f M, errors: %s" % (sys.argv[2], sys.argv[3], sys.argv[4], j["value"] / 1e6, list(bad) or "none"))
for k, v in bad.items():
    print("  ", k, v["error"]); print(v.get("stderr_tail", ""))
if bad: open("gpurun_out/r3l/FAILED_%s.json" % sys.argv[2], "w").write(json.dumps(j))
P
  if ls gpurun_out/r3l/FAILED_* >/dev/null 2>&1; then cp gpurun_out/r3l/reh.err gpurun_out/r3l/FAILED.err; break; fi
done
