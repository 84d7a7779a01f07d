// tools/clock_probe.hip - what clock does a LONE wavefront run at on this box, and when?
//
// The QC-latency half of the metric is one wavefront's serial chain: its time is instructions x cycles per
// instruction / shader clock, so two boxes whose lone-wavefront clock differs show exactly the "two classes of box" round 3
// could only guess at (key-cache QC p50 0.087 against 0.134 ms).  This probe measures that clock directly:
// one wavefront runs a dependent chain of v_mad_u64_u32 (the instruction the field arithmetic is made of) and reads
//   s_memtime      (clock64():      the shader-engine cycle counter), and
//   s_memrealtime  (wall_clock64(): a constant-rate counter, hipDeviceAttributeWallClockRate kHz)
// before and after; cycles / wall time = the effective clock of THAT wavefront while it ran.  Conditions:
//   cold      after the GPU has idled for `idle_ms` (default 200): does the power manager leave the clock low for a short kernel?
//   spaced    one launch every 1 ms (a node verifying a certificate per round)
//   back2back 20 launches in a row, the last one reported
//   loaded    while another stream keeps every CU busy with the same arithmetic (the throughput regime)
//   after     right after that load stopped
// Output: one line per condition - chain time (us, host-visible and from the counters), cycles, effective MHz, cycles per
// instruction.   hipcc --offload-arch=gfx950 -O2 tools/clock_probe.hip -o tools/clock_probe && tools/clock_probe [idle_ms]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <chrono>
#include <thread>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

#define CHAIN_ITERS 4096      // x 8 dependent multiply-adds = 32,768 instructions on the chain: ~0.1 - 0.2 ms, a QC's order of magnitude

__global__ void __launch_bounds__(64) k_chain(uint64_t *out, uint32_t seed) {
    uint64_t c = seed + threadIdx.x;
    const uint32_t a = seed * 2654435761u + threadIdx.x, m = seed | 1;
    const uint64_t w0 = wall_clock64();
    const uint64_t t0 = clock64();
    for (int it = 0; it < CHAIN_ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(m) : "vcc");
    }
    const uint64_t t1 = clock64();
    const uint64_t w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = c; }
}
// The ISSUE side of a lone wavefront: eight INDEPENDENT multiply-add accumulators (and eight independent 32-bit adds), so that
// nothing but the SIMD's willingness to issue from one wavefront bounds the rate - the dependent chain above measures latency,
// which two boxes can share while their issue cadence differs.  Launched as one wavefront on the whole chip, and as a workgroup
// of eight wavefronts (two per SIMD of one CU): cycles per instruction of ONE wavefront in each case.
__global__ void __launch_bounds__(512) k_indep(uint64_t *out, uint32_t seed, int what) {
    uint64_t c[8];
    uint32_t d[8];
    const uint32_t a = seed * 2654435761u + threadIdx.x, m = seed | 1;
    for (int i = 0; i < 8; i++) { c[i] = a + i; d[i] = a ^ i; }
    __syncthreads();
    const uint64_t t0 = clock64();
    if (what == 0) {
        for (int it = 0; it < CHAIN_ITERS; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(m) : "vcc");
        }
    } else {
        for (int it = 0; it < CHAIN_ITERS; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_add_u32 %0, %1, %0" : "+v"(d[i]) : "v"(m));
        }
    }
    const uint64_t t1 = clock64();
    uint64_t x = 0;
    for (int i = 0; i < 8; i++) x += c[i] + d[i];
    if ((threadIdx.x & 63) == 0) { out[8 + (threadIdx.x >> 6)] = t1 - t0; if (x == 0x1234567ull) out[7] = x; }
}
// the load: every SIMD busy with two wavefronts of the same instruction for a few milliseconds
__global__ void __launch_bounds__(256) k_load(uint64_t *out, uint32_t seed, int iters) {
    uint64_t c[4];
    const uint32_t a = seed + threadIdx.x, m = seed | 1;
    for (int i = 0; i < 4; i++) c[i] = a + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(m) : "vcc");
    }
    if (c[0] + c[1] + c[2] + c[3] == 12345) out[3] = 1;
}

struct sample { double host_us, wall_us, mhz, cpi; unsigned long long cycles; };

static sample run_chain(hipStream_t st, uint64_t *d_out, uint64_t *h_out, double wall_khz) {
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, st, d_out, 7u);
    CK(hipStreamSynchronize(st));
    const auto t1 = std::chrono::steady_clock::now();
    CK(hipMemcpy(h_out, d_out, 24, hipMemcpyDeviceToHost));
    sample s;
    s.host_us = std::chrono::duration<double, std::micro>(t1 - t0).count();
    s.cycles = h_out[0];
    s.wall_us = (double)h_out[1] / wall_khz * 1e3;
    s.mhz = s.wall_us > 0 ? (double)h_out[0] / s.wall_us : 0;
    s.cpi = (double)h_out[0] / (8.0 * CHAIN_ITERS);
    return s;
}
static void report(const char *tag, std::vector<sample> v) {
    std::sort(v.begin(), v.end(), [](const sample &a, const sample &b) { return a.wall_us < b.wall_us; });
    const sample &m = v[v.size() / 2], &lo = v.front(), &hi = v.back();
    printf("%-10s n=%-3zu chain p50 %8.1f us (min %.1f max %.1f; host-visible p50 %.1f)  cycles %llu  effective %7.1f MHz (min %.1f max %.1f)  %.2f cycles/instruction\n",
           tag, v.size(), m.wall_us, lo.wall_us, hi.wall_us, m.host_us, m.cycles, m.mhz, hi.mhz < lo.mhz ? hi.mhz : lo.mhz, hi.mhz > lo.mhz ? hi.mhz : lo.mhz, m.cpi);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int idle_ms = argc > 1 ? atoi(argv[1]) : 200;
    CK(hipSetDevice(0));
    int wall_khz = 0, clk_khz = 0, cus = 0;
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("device 0: %d CUs, reported max clock %d MHz, wall-clock counter %d kHz; chain = %d dependent v_mad_u64_u32 on one wavefront\n",
           cus, clk_khz / 1000, wall_khz, 8 * CHAIN_ITERS);
    hipStream_t st, st2;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    uint64_t *d_out, h_out[4];
    CK(hipMalloc(&d_out, 256));
    run_chain(st, d_out, h_out, wall_khz);       // module load, first launch
    std::vector<sample> v;
    for (int i = 0; i < 8; i++) { std::this_thread::sleep_for(std::chrono::milliseconds(idle_ms)); v.push_back(run_chain(st, d_out, h_out, wall_khz)); }
    report("cold", v); v.clear();
    for (int i = 0; i < 50; i++) { std::this_thread::sleep_for(std::chrono::milliseconds(1)); v.push_back(run_chain(st, d_out, h_out, wall_khz)); }
    report("spaced-1ms", v); v.clear();
    for (int i = 0; i < 50; i++) v.push_back(run_chain(st, d_out, h_out, wall_khz));
    report("back2back", v); v.clear();
    // under load: keep ~40 ms of k_load queued on the other stream while the chain is sampled
    for (int i = 0; i < 20; i++) {
        for (int k = 0; k < 4; k++) hipLaunchKernelGGL(k_load, dim3(cus * 8), dim3(256), 0, st2, d_out, 3u, 40000);
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        v.push_back(run_chain(st, d_out, h_out, wall_khz));
        CK(hipStreamSynchronize(st2));
    }
    report("loaded", v); v.clear();
    for (int i = 0; i < 20; i++) v.push_back(run_chain(st, d_out, h_out, wall_khz));
    report("after-load", v); v.clear();
    for (int i = 0; i < 4; i++) { std::this_thread::sleep_for(std::chrono::milliseconds(idle_ms)); v.push_back(run_chain(st, d_out, h_out, wall_khz)); }
    report("cold-again", v);
    // issue cadence of one wavefront: alone on the chip / one of two on its SIMD
    for (int what = 0; what < 2; what++) {
        for (int waves : {1, 8}) {
            std::vector<double> cpi;
            for (int rep = 0; rep < 9; rep++) {
                hipLaunchKernelGGL(k_indep, dim3(1), dim3(64 * waves), 0, st, d_out, 7u, what);
                CK(hipStreamSynchronize(st));
                uint64_t h[16];
                CK(hipMemcpy(h, d_out, 128, hipMemcpyDeviceToHost));
                double worst = 0;
                for (int w = 0; w < waves; w++) worst = std::max(worst, (double)h[8 + w] / (8.0 * CHAIN_ITERS));
                cpi.push_back(worst);
            }
            std::sort(cpi.begin(), cpi.end());
            printf("issue      %-14s independent x8, %d wavefront%s in one workgroup (%s): %.2f cycles per instruction of one wavefront (min %.2f max %.2f)\n",
                   what == 0 ? "v_mad_u64_u32" : "v_add_u32", waves, waves > 1 ? "s" : "", waves > 1 ? "two per SIMD" : "alone on the chip", cpi[4], cpi[0], cpi[8]);
        }
    }
    return 0;
}
