#!/bin/bash
# Round-4 gpurun visits.  gpurun --timeout 1500 -- 'bash tools/gpu_r4.sh <stage>'   (everything lands in gpurun_out/r4_<stage>_*)
stage=${1:-a}
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_${stage}
env_record() {      # clocks / power / partitions / the lone-wavefront clock, beside whatever is measured next
    python tools/gpu_env.py > ${O}_env_$1.json 2>${O}_env_$1.err
    rocm-smi --showclocks --showperflevel --showpower --showmaxpower --showcomputepartition --showmemorypartition --showsclkrange > ${O}_rocm_smi_$1.txt 2>&1
    timeout 60 tools/clock_probe 200 > ${O}_clock_probe_$1.txt 2>&1
    grep -E "cold|spaced|back2back|loaded|after" ${O}_clock_probe_$1.txt | cut -c1-170
}
case $stage in
a)  # tests of the round's first changes + the idle-context coupling probe under three hardware-queue budgets + the driver's bench
    env_record start
    timeout 900 python -m pytest tests -x -q -m gpu > ${O}_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 ${O}_pytest.txt
    for q in 16 8 24; do
        GPU_MAX_HW_QUEUES=$q timeout 300 python tools/streamed_probe.py coupling > ${O}_coupling_q$q.txt 2>&1; echo "coupling q=$q rc=$?"
        grep -E "in flight" ${O}_coupling_q$q.txt | cut -c1-150
    done
    timeout 600 python bench.py --steps 20 --warmup 5 > ${O}_bench_default.json.txt 2>${O}_bench_default.err; echo "bench rc=$?"
    python - <<'PY'
import json
j = json.loads(open("gpurun_out/r4_a_bench_default.json.txt").read().strip().splitlines()[-1])
q = j.get("qc_verify_latency", {})
print("headline %.2f M  chip_frac %s issue_frac %s | qc cold %s cached %s | env sclk %s lone-wave %s" % (
    j["value"] / 1e6, j["roofline"].get("chip_frac"), (j.get("int_roofline") or {}).get("issue_frac"),
    q.get("p10_p50_p99_ms_device_resident"), q.get("committee_cached", {}).get("p10_p50_p99_ms_device_resident"),
    (q.get("env", {}).get("rocm_smi") or {}).get("sclk"), q.get("env", {}).get("lone_wavefront_clock")))
print("multi_process", json.dumps(j.get("multi_process"))[:600])
PY
    env_record end
    ;;
esac
