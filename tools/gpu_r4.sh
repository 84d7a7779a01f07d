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
    grep -E "cold|spaced|back2back|loaded|after|issue" ${O}_clock_probe_$1.txt | cut -c1-170
    timeout 120 tools/mem_latency_probe > ${O}_mem_latency_$1.txt 2>&1; grep -v independent ${O}_mem_latency_$1.txt
    timeout 60 tools/placement_probe > ${O}_placement_$1.txt 2>&1; head -3 ${O}_placement_$1.txt | cut -c1-330
    timeout 60 tools/ifetch_probe > ${O}_ifetch_$1.txt 2>&1; cat ${O}_ifetch_$1.txt | cut -c1-200
    timeout 200 python tools/cached_trace.py > ${O}_cached_trace_$1.md 2>&1; cat ${O}_cached_trace_$1.md | cut -c1-300
    timeout 120 python tools/qc_marker.py > ${O}_qc_marker_$1.json 2>${O}_qc_marker_$1.err; echo "class marker: $(cat ${O}_qc_marker_$1.json)"
}
pmc_record() {      # SQ / instruction-cache counters of the two QC kernels (lone calls of 6,667 signatures), two passes
    for pass in sq:"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_IFETCH SQ_WAVES GRBM_GUI_ACTIVE" \
                sqc:"SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_MISSES"; do
        name=${pass%%:*}; ctrs=${pass#*:}
        rm -rf /tmp/pmc_$name
        timeout 200 rocprofv3 --pmc $ctrs -d /tmp/pmc_$name -o p -- python tools/qc_marker.py > ${O}_pmc_${name}_$1.marker.json 2> ${O}_pmc_${name}_$1.err; echo "pmc $name rc=$?"
        db=$(find /tmp/pmc_$name -name "*results.db" | head -1)
        [ -n "$db" ] && python tools/rocpd_pmc.py $db --by-grid k_verify_cached_lane2r k_verify_fused8 > ${O}_pmc_${name}_$1.json && python - ${O}_pmc_${name}_$1.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
for k, v in sorted(j.items()):
    print(k, {c: round(x["avg_per_dispatch"]) for c, x in sorted(v.items())})
PY
    done
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
b)  # the rest of the suite after visit a's one failure, the four-tenant sweep, the sanitizer runs
    env_record start
    timeout 900 python -m pytest tests -q -m gpu > ${O}_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 ${O}_pytest.txt
    timeout 600 python tools/mp_sweep.py 0.8 > ${O}_mp_sweep.txt 2>${O}_mp_sweep.err; echo "mp_sweep rc=$?"; cat ${O}_mp_sweep.txt | cut -c1-230
    for kind in tsan asan; do
        d=build/san/$kind
        [ -x $d/stress_host ] || { echo "no $kind build"; continue; }
        if [ $kind = tsan ]; then export TSAN_OPTIONS="suppressions=$PWD/tests/cpp/tsan.supp history_size=4 second_deadlock_stack=1 exitcode=66 log_path=$PWD/${O}_tsan_report"
        else export ASAN_OPTIONS="detect_leaks=0 exitcode=67 log_path=$PWD/${O}_asan_report protect_shadow_gap=0" UBSAN_OPTIONS="print_stacktrace=1"; fi
        timeout 400 $d/stress_host 6 > ${O}_${kind}_stress.txt 2>&1; echo "$kind stress_host rc=$?"; tail -2 ${O}_${kind}_stress.txt
        PYTHONPATH=tools:tests python tests/test_gpu_cpp_mirror.py fixture /tmp/mirror_fixture.bin && timeout 200 $d/mirror_tests /tmp/mirror_fixture.bin > ${O}_${kind}_mirror.txt 2>&1; echo "$kind mirror_tests rc=$?"; grep -c "^ok" ${O}_${kind}_mirror.txt; grep "^FAIL" ${O}_${kind}_mirror.txt | head
        unset TSAN_OPTIONS ASAN_OPTIONS UBSAN_OPTIONS
        ls ${O}_${kind}_report* 2>/dev/null | head -5
        grep -h "SUMMARY" ${O}_${kind}_report* 2>/dev/null | sort | uniq -c | sort -rn | head -20
    done
    ;;
c)  # the split eight-lane kernel A/B, the queue-oversubscription probe, sanitizer canaries, the suite, the driver's bench
    env_record start
    timeout 600 python tools/oct_split_ab.py > ${O}_oct_split_ab.txt 2>${O}_oct_split_ab.err; echo "oct_split_ab rc=$?"; cat ${O}_oct_split_ab.txt
    timeout 900 python -m pytest tests -x -q -m gpu > ${O}_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 ${O}_pytest.txt
    timeout 600 python tools/queue_probe.py 0.6 > ${O}_queue_probe.txt 2>${O}_queue_probe.err; echo "queue_probe rc=$?"; cat ${O}_queue_probe.txt | cut -c1-200
    for kind in tsan asan; do
        d=build/san/$kind
        if [ $kind = tsan ]; then export TSAN_OPTIONS="suppressions=$PWD/tests/cpp/tsan.supp exitcode=66"; else export ASAN_OPTIONS="detect_leaks=0 exitcode=67"; fi
        timeout 60 $d/stress_host canary > ${O}_${kind}_canary.txt 2>&1; echo "$kind canary rc=$? (66 / 67 = the detector fired)"; grep -E "SUMMARY|WARNING: ThreadSanitizer|ERROR: AddressSanitizer" ${O}_${kind}_canary.txt | head -3
        timeout 400 $d/stress_host 6 > ${O}_${kind}_stress.txt 2>&1; echo "$kind stress_host rc=$?"; tail -1 ${O}_${kind}_stress.txt
        unset TSAN_OPTIONS ASAN_OPTIONS
    done
    timeout 600 python bench.py --steps 20 --warmup 5 > ${O}_bench_default.json.txt 2>${O}_bench_default.err; echo "bench rc=$?"
    python - <<'PY'
import json
j = json.loads(open("gpurun_out/r4_c_bench_default.json.txt").read().strip().splitlines()[-1])
q = j.get("qc_verify_latency", {})
print("headline %.2f M | qc cold %s cached %s | kfd %s" % (j["value"] / 1e6, q.get("p10_p50_p99_ms_device_resident"),
      q.get("committee_cached", {}).get("p10_p50_p99_ms_device_resident"), q.get("env", {}).get("kfd")))
for r in j.get("multi_process", {}).get("rows", []): print(r)
PY
    ;;
d)  # memory latency beside the clocks; the suite (sharded one-combination tests); streams-sharing-queues probe; tenants with
    # the smaller pools: sweep + the idle-holder probe; the driver's bench
    env_record start
    timeout 900 python -m pytest tests -x -q -m gpu > ${O}_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 ${O}_pytest.txt
    timeout 400 python tools/streamed_probe.py coupling2 > ${O}_coupling2.txt 2>&1; echo "coupling2 rc=$?"; grep "in flight" ${O}_coupling2.txt | cut -c1-160
    timeout 600 python tools/queue_probe.py 0.6 > ${O}_queue_probe.txt 2>${O}_queue_probe.err; echo "queue_probe rc=$?"; cat ${O}_queue_probe.txt | cut -c1-200
    timeout 600 python bench.py --steps 20 --warmup 5 > ${O}_bench_default.json.txt 2>${O}_bench_default.err; echo "bench rc=$?"
    python - <<'PY'
import json
j = json.loads(open("gpurun_out/r4_d_bench_default.json.txt").read().strip().splitlines()[-1])
q = j.get("qc_verify_latency", {})
print("headline %.2f M | qc cold %s kernel %s | cached %s kernel %s" % (j["value"] / 1e6, q.get("p10_p50_p99_ms_device_resident"), q.get("kernel_ms_device_resident"),
      q.get("committee_cached", {}).get("p10_p50_p99_ms_device_resident"), q.get("committee_cached", {}).get("kernel_ms_device_resident")))
print("gather", q.get("env", {}).get("dependent_gather"))
for r in j.get("multi_process", {}).get("rows", []): print(r)
PY
    ;;
e)  # box identity + gather throughput beside the clocks; key-table footprint probe; 3 against 4 wavefronts per SIMD for the
    # key-cache kernel; tenants with the pools divided like the single process's; the suite
    env_record start
    timeout 300 python tools/cached_footprint_probe.py > ${O}_cached_footprint.md 2>${O}_cached_footprint.err; echo "footprint rc=$?"; cat ${O}_cached_footprint.md
    timeout 500 python tools/cached_waves_ab.py > ${O}_cached_waves_ab.md 2>${O}_cached_waves_ab.err; echo "cached_waves_ab rc=$?"; cat ${O}_cached_waves_ab.md | cut -c1-220
    timeout 600 python tools/mp_sweep.py 0.6 > ${O}_mp_sweep.txt 2>${O}_mp_sweep.err; echo "mp_sweep rc=$?"; cat ${O}_mp_sweep.txt | cut -c1-230
    timeout 900 python -m pytest tests -x -q -m gpu > ${O}_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 ${O}_pytest.txt
    ;;
f)  # inputs staged through LDS (coalesced, every byte once) against per-lane record loads: the suite, then the same-box A/B
    # (the HS_ED25519_PREPARE_STAGED / HS_ED25519_CACHED_STAGED switches exist at commit 8f5d7a8 only; the winners are built in since)
    timeout 900 python -m pytest tests -x -q -m gpu > ${O}_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 ${O}_pytest.txt
    timeout 60 tools/ifetch_probe > ${O}_ifetch.txt 2>&1; head -4 ${O}_ifetch.txt | cut -c1-200
    timeout 1200 python tools/staged_ab.py > ${O}_staged_ab.md 2>${O}_staged_ab.err; echo "staged_ab rc=$?"; cat ${O}_staged_ab.md | cut -c1-700
    ;;
g)  # confirmation at the simplified kernels (staged batch lists, direct lone key-cache call): suite, smoke, rocprofv3 kernel stats of
    # the headline, the driver's bench; then the host path against the narrow pool's shape (8 / 4 / 2 slices) with today's kernels
    timeout 900 python -m pytest tests -x -q -m gpu > ${O}_pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" ${O}_pytest.txt | tail -3
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
    timeout 60 tools/ifetch_probe > ${O}_ifetch.txt 2>&1; sed -n 2,3p ${O}_ifetch.txt | cut -c1-120
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kth -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu --no-extras > ${O}_bench_headline_under_rocprof.json.txt 2> ${O}_kth.err; echo "rocprof rc=$?"
    python tools/rocpd_summary.py $(find /tmp/kth -name "*results.db" | head -1) > ${O}_bench_headline_kernel_stats.csv; head -4 ${O}_bench_headline_kernel_stats.csv | cut -c1-160
    timeout 600 python bench.py --steps 20 --warmup 5 > ${O}_bench_default.json.txt 2>${O}_bench_default.err; echo "bench rc=$?"
    python - <<'PY'
import json
j = json.loads(open("gpurun_out/r4_g_bench_default.json.txt").read().strip().splitlines()[-1])
q = j.get("qc_verify_latency", {})
print("headline %.2f M | kernels %s | qc cold %s kernel %s | cached %s kernel %s | e2e %.1f M" % (j["value"] / 1e6, j["kernels_ms"], q.get("p10_p50_p99_ms_device_resident"), q.get("kernel_ms_device_resident"),
      q.get("committee_cached", {}).get("p10_p50_p99_ms_device_resident"), q.get("committee_cached", {}).get("kernel_ms_device_resident"), j["end_to_end"]["value"] / 1e6))
print("cached", j["committee_cached"]["value"] / 1e6, "streamed", json.dumps(j.get("streamed_rounds"))[:300])
for r in j.get("multi_process", {}).get("rows", []): print(r)
PY
    for P in 8 4 2; do
        HS_ED25519_POOL_PARTITIONS=$P timeout 200 python tools/e2e_async.py 16x16x16 4x32x16 1x256x16 16x8x8 16x4x4 16x2x2 > ${O}_e2e_shape_p$P.txt 2>&1; cat ${O}_e2e_shape_p$P.txt | cut -c1-150
    done
    HS_ED25519_POOL_PARTITIONS=4 HS_ED25519_POOL_NARROW=8 timeout 200 python tools/e2e_async.py 16x16x16 1x256x16 16x8x8 > ${O}_e2e_shape_p4n8.txt 2>&1; cat ${O}_e2e_shape_p4n8.txt | cut -c1-150
    for P in 8 4; do
        timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-extras --partitions $P 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('device-resident, sixteen batches per launch, $P partitions: %.2f M  kernels %s' % (j['value'] / 1e6, j['kernels_ms']))"
    done
    ;;
j)  # key cache: chains + batched finish by compression (the "c" form) - parity first, then the three forms by size and the batch lists
    timeout 60 tools/ifetch_probe > ${O}_ifetch.txt 2>&1; sed -n 2,3p ${O}_ifetch.txt | cut -c1-120
    timeout 900 python -m pytest tests/test_gpu_keyset.py tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py tests/test_gpu_ring.py tests/test_gpu_rlc.py -x -q -m gpu > ${O}_pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" ${O}_pytest.txt | tail -3; grep -E "^(FAILED|ERROR|E  )" ${O}_pytest.txt | head -20
    timeout 1200 python tools/cached_c_ab.py > ${O}_cached_c_ab.md 2>${O}_cached_c_ab.err; echo "cached_c_ab rc=$?"; cat ${O}_cached_c_ab.md | cut -c1-400
    ;;
o)  # the round's last confirmation: suite, smoke, the driver's bench
    timeout 60 tools/ifetch_probe > ${O}_ifetch.txt 2>&1; sed -n 2,2p ${O}_ifetch.txt | cut -c1-100
    timeout 1200 python -m pytest tests -x -q -m gpu > ${O}_pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" ${O}_pytest.txt | tail -3
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
    timeout 600 python bench.py --steps 20 --warmup 5 > ${O}_bench_default.json.txt 2>${O}_bench_default.err; echo "bench rc=$?"
    python - <<'PY'
import json
j = json.loads(open("gpurun_out/r4_o_bench_default.json.txt").read().strip().splitlines()[-1])
q = j.get("qc_verify_latency", {})
print("headline %.2f M | kernels %s | qc cold %s kernel %s (%s) | cached %s kernel %s | e2e %.1f M" % (j["value"] / 1e6, j["kernels_ms"], q.get("p10_p50_p99_ms_device_resident"), q.get("kernel_ms_device_resident"), q.get("kernel"),
      q.get("committee_cached", {}).get("p10_p50_p99_ms_device_resident"), q.get("committee_cached", {}).get("kernel_ms_device_resident"), j["end_to_end"]["value"] / 1e6))
cc = j["committee_cached"]
print("cached %.1f M (%s, %.4f ms)  large committee %s" % (cc["value"] / 1e6, cc["kernel"], cc["avg_kernel_ms"], {k: cc["large_committee"].get(k) for k in ("kernel", "avg_kernel_ms", "value")}))
print("roofline", {k: j["roofline"].get(k) for k in ("frac", "chip_frac", "achieved", "traffic", "avg_kernel_ms")}, "int", {k: (j.get("int_roofline") or {}).get(k) for k in ("frac", "issue_frac")})
print("cpu", {k: j["cpu_baseline"].get(k) for k in ("value", "cores", "kind")}, "rlc_ab", json.dumps(j.get("rlc_ab"))[:300])
for r in j.get("multi_process", {}).get("rows", []): print(r)
PY
    ;;
p)  # round-4 evidence beside the headline: SQ counters of the launch pair, kernel stats of the FULL default command, streamed rounds for 10 s
    timeout 60 tools/ifetch_probe > ${O}_ifetch.txt 2>&1; sed -n 2,2p ${O}_ifetch.txt | cut -c1-100
    rm -rf /tmp/pmc_sq; timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d /tmp/pmc_sq -o p -- python bench.py --steps 3 --warmup 1 --no-extras --no-cpu > /dev/null 2> ${O}_pmc_sq.err; echo "pmc sq rc=$?"
    python tools/rocpd_pmc.py $(find /tmp/pmc_sq -name "*results.db" | head -1) k_verify_batches k_prepare_batches > ${O}_pmc_sq_headline.json; python - <<'PY'
import json
j = json.load(open("gpurun_out/r4_p_pmc_sq_headline.json"))
tot = 0
for k, v in sorted(j.items()):
    per = v["SQ_INSTS_VALU"]["avg_per_dispatch"] / 16384
    tot += per
    print(k, "VALU wave-instructions per verification %.1f" % per, {c: round(x["avg_per_dispatch"]) for c, x in v.items() if c in ("SQ_WAVES", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY")})
print("pair: %.1f per verification" % tot)
PY
    rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu > ${O}_bench_under_rocprof.json.txt 2> ${O}_kt.err; echo "rocprof full rc=$?"
    python tools/rocpd_summary.py $(find /tmp/kt -name "*results.db" | head -1) > ${O}_bench_default_kernel_stats.csv; head -14 ${O}_bench_default_kernel_stats.csv | cut -c1-60,150-
    timeout 200 python tools/streamed_10s.py 10 > ${O}_streamed_10s.md 2>${O}_streamed.err; echo "streamed rc=$?"; cat ${O}_streamed_10s.md
    ;;
s)  # sanitizer runs at the round's last code (host side built with -fsanitize=thread / address,undefined; __graft_entry__.build_sanitized)
    for kind in tsan asan; do
        d=build/san/$kind
        [ -x $d/stress_host ] || { echo "no $kind build"; continue; }
        if [ $kind = tsan ]; then export TSAN_OPTIONS="suppressions=$PWD/tests/cpp/tsan.supp history_size=4 second_deadlock_stack=1 exitcode=66 log_path=$PWD/${O}_tsan_report"
        else export ASAN_OPTIONS="detect_leaks=0 exitcode=67 log_path=$PWD/${O}_asan_report protect_shadow_gap=0" UBSAN_OPTIONS="print_stacktrace=1"; fi
        timeout 60 $d/stress_host canary > ${O}_${kind}_canary.txt 2>&1; echo "$kind canary rc=$? (66 / 67 = the detector fired)"
        rm -f ${O}_${kind}_report*
        timeout 300 $d/stress_host 6 > ${O}_${kind}_stress.txt 2>&1; echo "$kind stress_host rc=$?"; tail -2 ${O}_${kind}_stress.txt
        PYTHONPATH=tools:tests python tests/test_gpu_cpp_mirror.py fixture /tmp/mirror_fixture.bin && timeout 200 $d/mirror_tests /tmp/mirror_fixture.bin > ${O}_${kind}_mirror.txt 2>&1; echo "$kind mirror_tests rc=$?"; grep -c "^ok" ${O}_${kind}_mirror.txt; grep "^FAIL" ${O}_${kind}_mirror.txt | head
        unset TSAN_OPTIONS ASAN_OPTIONS UBSAN_OPTIONS
        ls ${O}_${kind}_report* 2>/dev/null | head -5
        grep -h "SUMMARY" ${O}_${kind}_report* 2>/dev/null | sort | uniq -c | sort -rn | head -10
    done
    ;;
env)  # what box is this: clocks, identity, memory latency, wavefront placement - nothing else
    env_record only
    ;;
pmc)  # the QC kernels' SQ / instruction-cache counters beside the class marker (two rocprofv3 passes)
    env_record only
    pmc_record only
    ;;
esac
