#!/bin/bash
# Round-3 GPU visit C: why the half-size form gains less than its operation count says.
mkdir -p gpurun_out/r3
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3
echo "== half_probe (GPU vs host)"; timeout 120 tools/half_probe
echo "== rehearsal --gpus 2"; timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --batches-per-step 64 --large-committee 3000 > $O/rehearsal_c.json.txt 2> $O/rehearsal_c.err; echo "rc=$?"
grep -E "^\[rank|^bench:|Error" $O/rehearsal_c.err | head -40
echo "== latency, shipped library vs forced 36 windows"
for lib in hotstuff_amd/libhs_ed25519.so build/libhs_force36.so; do
  echo "-- $lib"
  HS_ED25519_LIB=$GRAFT_REPO_ROOT/$lib HS_SWEEP_SIZES=1024,6667,16384,66667 timeout 300 python tools/r2_sweeps.py 2>/dev/null | grep -E "^\| [0-9]"
done
echo "== headline, shipped vs forced 36 windows"
for lib in hotstuff_amd/libhs_ed25519.so build/libhs_force36.so; do
  HS_ED25519_LIB=$GRAFT_REPO_ROOT/$lib timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('$lib: %.2f M  %s %.4f ms prepare %s' % (j['value']/1e6, j['roofline']['kernel'], j['roofline']['avg_kernel_ms'], j['kernels_ms']['k_prepare']))"
done
echo "== full GPU suite (no -x)"; timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu_c.txt 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/pytest_gpu_c.txt | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_gpu_c.txt | head -20
