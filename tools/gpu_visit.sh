#!/bin/bash
# gpurun visits (rounds 5 and 6).  gpurun --timeout 1500 -- 'bash tools/gpu_visit.sh <tag> <stage> [stage ...]'   (files land in gpurun_out/r6_<tag>_*)
#   tests   the GPU suite + smoke()
#   bench   the driver's command; the line, its detail file and a one-screen summary
#   diff    the scale differential: 10^7 crafted tuples through every kernel form + 10^7 / 10^5 scalars through half_scalar
#   prof    rocprofv3 --kernel-trace --stats of the headline workload (the kernel's average duration the roofline must agree with)
#   stream  BASELINE configs[4]: 10 s of back-to-back 6,667-signature rounds from one host thread through the ring
#   cap     tools/capture_probe (which runtime calls of another thread break a thread-local capture) + the ring / key-set
#           thread tests ten times over
#   e2e     the host-API rate (tools/e2e_async.py) against the narrow pool's shape: HS_ED25519_POOL_PARTITIONS x HS_ED25519_POOL_DEPTH, one
#           process per shape (the knobs are read at init); E2E_SHAPES / E2E_CONFIGS override the rows
#   san     the host side under -fsanitize=thread and address,undefined (python -c "import __graft_entry__ as g; g.build_sanitized()"
#           first, here: build/san/ travels with the snapshot): canary, the 15-thread stress client, the C++ mirror tests
tag=${1:-a}; shift
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${HS_ROUND:-r6}_${tag}
for stage in "$@"; do
case $stage in
tests)
    timeout 1500 python -m pytest tests -x -q -m gpu > ${O}_pytest.txt 2>&1; echo "pytest rc=$?"
    grep -E "passed|failed|error" ${O}_pytest.txt | tail -3
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
    ;;
bench)
    timeout 900 python bench.py --steps 20 --warmup 5 --detail-file ${O}_bench_detail.json > ${O}_bench.json.txt 2> ${O}_bench.err; echo "bench rc=$? line bytes $(wc -c < ${O}_bench.json.txt)"
    python - ${O}_bench.json.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: j = json.loads(l)
    except Exception: continue
    c, r = j["config"], j["roofline"]
    g = lambda k: c.get(k)
    print("headline %.2f M verifies/s  %s %.4f ms  traffic %s (x%s algorithmic)  mad %s issue %s" % (j["value"] / 1e6, r["kernel"], r["avg_kernel_ms"], r["traffic"], r.get("traffic_over_algorithmic"), g("int_roofline_mad_frac"), g("int_roofline_issue_frac")))
    print("QC N=%s: cold p50 %s ms (kernel %s, %s by %s; fused8 %s / fused8h %s ms here, autotune %s / %s), key cache p50 %s ms (kernel %s); host API %s / %s ms" % (
        g("qc_validators"), g("qc_verify_p50_ms_cold"), g("qc_kernel_ms_cold"), g("qc_cold_form_chosen"), g("qc_cold_form_chosen_by"), g("qc_kernel_ms_verify_fused8"), g("qc_kernel_ms_verify_fused8h"),
        g("autotune_ms_verify_fused8"), g("autotune_ms_verify_fused8h"), g("qc_verify_p50_ms_key_cache"), g("qc_kernel_ms_key_cache"), g("qc_verify_p50_ms_cold_host_api"), g("qc_verify_p50_ms_key_cache_host_api")))
    print("host->host best %s (%s)  one thread %s (%s)  16x8 %s | one sync caller 1,024: %s (%s ms)  QC cold %s  QC cached %s | 4 tenants %s (spread %s), bursts of 4: %s" % (
        j.get("value_host_to_host"), g("end_to_end_best_shape"), g("e2e_one_thread_tickets_verifies_per_s"), g("e2e_one_thread_tickets_shape"), g("e2e_16_threads_x_8_tickets_verifies_s"),
        g("one_sync_caller_1024_verifies_per_s"), g("one_sync_caller_1024_call_p50_ms"), g("one_sync_caller_verifies_per_s_qc_cold"), g("one_sync_caller_verifies_per_s_qc_cached"),
        g("four_tenants_verifies_per_s"), g("four_tenants_max_over_min"), g("four_tenants_bursts_of_4_verifies_per_s")))
    print("TC N=%s: cold p50 %s ms (kernel %s), key cache %s (kernel %s); host API %s / %s | CPU twins (%s cores): QC verify_batch %s / %s ms, TC loop %s / %s ms" % (
        g("tc_validators"), g("tc_verify_p50_ms_cold"), g("tc_kernel_ms_cold"), g("tc_verify_p50_ms_key_cache"), g("tc_kernel_ms_key_cache"), g("tc_verify_p50_ms_cold_host_api"),
        g("tc_verify_p50_ms_key_cache_host_api"), g("cpu_cores"), g("cpu_qc_verify_batch_p50_ms_one_core"), g("cpu_qc_verify_batch_p50_ms_all_cores"),
        g("cpu_tc_verify_p50_ms_one_core"), g("cpu_tc_verify_p50_ms_all_cores")))
    print("QC alternating %s / %s | streamed %s / %s | cached list %s, 66,667: %s ms frac %s | lane2r states/wave_cycles %s wait_any %s wait_inst %s active %s of %s" % (
        g("qc_verify_p50_ms_cold_alternating"), g("qc_verify_p50_ms_key_cache_alternating"), g("streamed_cold_signatures_per_s"), g("streamed_key_cache_signatures_per_s"),
        g("committee_cached_verifies_per_s"), g("cached_qc_66667_kernel_ms"), g("cached_qc_66667_hbm_frac"), g("sq_verify_cached_lane2r_states_over_wave_cycles"),
        g("sq_verify_cached_lane2r_wait_any"), g("sq_verify_cached_lane2r_wait_inst_any"), g("sq_verify_cached_lane2r_active_any"), g("sq_verify_cached_lane2r_wave_cycles")))
    print("callers either side, N = 10,000 (C++ twin): vote ingest -> QC %s ms in %s engine call(s) (reference, derived: %s ms on one core) | 200 timeouts with one high_qc %s ms, %s QC verification(s) (reference, derived: %s ms)" % (
        g("vote_ingest_qc_ms"), g("vote_ingest_engine_calls"), g("cpu_vote_ingest_ms_derived"), g("timeout_storm_200_ms"), g("timeout_storm_qc_verifies"), g("cpu_timeout_storm_200_ms_derived")))
    print("first 24 config keys:", list(c)[:24])
    print("cpu", j.get("cpu_baseline", {}).get("value"), "detail", j.get("detail_file"), "config keys", len(c))
PY
    tail -3 ${O}_bench.err | cut -c1-300
    ;;
diff)
    timeout 1200 python tools/differential.py --total ${HS_DIFF_TOTAL:-10000000} --out ${O}_differential.json 2> ${O}_differential.err; echo "differential rc=$?"
    tail -22 ${O}_differential.err | cut -c1-200
    timeout 600 tools/half_differential ${HS_HALF_TOTAL:-10000000} 100000 > ${O}_half_differential.json 2> ${O}_half_differential.err; echo "half_differential rc=$?"
    python -c "
import json
d = json.load(open('${O}_half_differential.json'))
for k in ('random', 'adversarial'):
    print(k, {x: v for x, v in d[k].items() if 'hist' not in x})
"
    ;;
prof)
    timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/r6_kth -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu --no-extras > ${O}_bench_headline_under_rocprof.json.txt 2> ${O}_kth.err; echo "rocprofv3 rc=$?"
    python tools/rocpd_summary.py $(find /tmp/r6_kth -name "*results.db" | head -1) > ${O}_bench_headline_kernel_stats.csv; head -4 ${O}_bench_headline_kernel_stats.csv | cut -c1-160
    ;;
stream)
    timeout 120 python tools/streamed_10s.py 10 > ${O}_streamed_10s.md 2> ${O}_streamed_10s.err; echo "streamed_10s rc=$?"; cat ${O}_streamed_10s.md
    ;;
cap)
    [ -x tools/capture_probe ] || hipcc --offload-arch=gfx950 -O2 -o tools/capture_probe tools/capture_probe.hip -lpthread
    timeout 120 tools/capture_probe 1500 > ${O}_capture_probe.json 2> ${O}_capture_probe.err; echo "capture_probe rc=$?"
    python - ${O}_capture_probe.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("HIP runtime", d["hip_runtime_version"])
for c in d["candidates"]:
    print("%-48s calls %8d errs %-40s captures ok %6d failed %5d %s" % (c["other_thread"], c["its_calls"], c["its_errors"], c["captures_ok"], c["captures_failed"], c["failures"]))
PY
    for i in 1 2 3 4 5 6 7 8 9 10; do
        timeout 300 python -m pytest tests/test_gpu_ring.py tests/test_gpu_keyset.py -x -q -m gpu -s -k "captures or synchronises or destroy_waits" > ${O}_cap_pytest_$i.txt 2>&1
        echo "run $i rc=$? $(grep -E 'passed|failed' ${O}_cap_pytest_$i.txt | tail -1) $(grep -h 'device synchronisations' ${O}_cap_pytest_$i.txt)"
    done
    ;;
e2e)
    : > ${O}_e2e_pool_shape.md
    for cfg in ${E2E_CONFIGS:-8:1 8:2 4:1 4:2}; do
        HS_ED25519_POOL_PARTITIONS=${cfg%%:*} HS_ED25519_POOL_DEPTH=${cfg##*:} timeout 300 python tools/e2e_async.py ${E2E_SHAPES:-1x1x1 16x1x1 16x2x2 16x4x4 16x8x8 16x16x16 4x32x16 4x64x16 1x256x16 1x512x16} >> ${O}_e2e_pool_shape.md 2>> ${O}_e2e.err
        echo >> ${O}_e2e_pool_shape.md
    done
    cat ${O}_e2e_pool_shape.md
    ;;
san)
    for kind in tsan asan; do
        d=build/san/$kind
        [ -x $d/stress_host ] || { echo "no $kind build"; continue; }
        if [ $kind = tsan ]; then export TSAN_OPTIONS="suppressions=$PWD/tests/cpp/tsan.supp history_size=4 second_deadlock_stack=1 exitcode=66 log_path=$PWD/${O}_tsan_report"
        else export ASAN_OPTIONS="detect_leaks=0 exitcode=67 log_path=$PWD/${O}_asan_report protect_shadow_gap=0 quarantine_size_mb=1024" UBSAN_OPTIONS="print_stacktrace=1" HS_TEST_QUICK_EXIT=1; fi      # (quick exit: ROCm's ASan fails its own CHECK inside the HIP runtime's exit finalizer, tests/cpp/stress_host.cpp)
        timeout 60 $d/stress_host canary > ${O}_${kind}_canary.txt 2>&1; echo "$kind canary rc=$? (66 / 67 = the detector fired)"
        rm -f ${O}_${kind}_report*
        timeout 600 $d/stress_host ${HS_STRESS_SECONDS:-6} > ${O}_${kind}_stress.txt 2>&1; echo "$kind stress_host rc=$?"; tail -2 ${O}_${kind}_stress.txt
        PYTHONPATH=tools:tests python tests/test_gpu_cpp_mirror.py fixture /tmp/mirror_fixture.bin && timeout 200 $d/mirror_tests /tmp/mirror_fixture.bin > ${O}_${kind}_mirror.txt 2>&1; echo "$kind mirror_tests rc=$?"; grep -c "^ok" ${O}_${kind}_mirror.txt; grep "^FAIL" ${O}_${kind}_mirror.txt | head
        unset TSAN_OPTIONS ASAN_OPTIONS UBSAN_OPTIONS HS_TEST_QUICK_EXIT
        ls ${O}_${kind}_report* 2>/dev/null | head -5
        grep -h "SUMMARY" ${O}_${kind}_report* 2>/dev/null | sort | uniq -c | sort -rn | head -10
    done
    ;;
esac
done
