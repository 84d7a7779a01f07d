// tools/ifetch_probe.hip - how fast does this box deliver INSTRUCTIONS to a wavefront?
//
// Round 4's per-phase trace of the key-cache QC (tools/cached_trace.py) showed pure register arithmetic - the R wavefront's
// square-root chain, the hash - running 1.2x slower on some boxes with ONE workgroup on the chip and 1.8x slower with 105,
// while every clock, issue-cadence, memory-latency and placement probe agreed between the boxes; the SQ counters of the slow
// boxes show the waves parked (SQ_WAIT_ANY 4.4x) and 60x more instruction-cache requests landing on a miss that is still
// pending (SQC_ICACHE_MISSES_DUPLICATE) for the same number of misses: instruction lines take longer to arrive.  This probe
// isolates that path: one wavefront per workgroup executes straight-line code (8-byte v_add_u32 with a literal) of
//   24 KB   - loops inside the 64 KB instruction cache after the first pass,
//   192 KB  - streams through it: every pass misses, lines come from the L2 / memory,
// eight passes each, on 1, 105 and 256 workgroups.  Reported: nanoseconds per instruction of the slowest wavefront.
//   hipcc --offload-arch=gfx950 -O2 tools/ifetch_probe.hip -o tools/ifetch_probe && tools/ifetch_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define PASSES 8

__global__ void __launch_bounds__(64) k_small(uint64_t *out, uint32_t seed) {       // 3,072 instructions = 24 KB
    uint32_t x = seed + threadIdx.x;
    const uint64_t t0 = wall_clock64();
    for (int p = 0; p < PASSES; p++) asm volatile(".rept 3072\n v_add_u32 %0, 0x12345, %0\n .endr" : "+v"(x));
    const uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (x == 0xdeadbeefu) out[0] = x;
}
__global__ void __launch_bounds__(64) k_large(uint64_t *out, uint32_t seed) {       // 24,576 instructions = 192 KB
    uint32_t x = seed + threadIdx.x;
    const uint64_t t0 = wall_clock64();
    for (int p = 0; p < PASSES; p++) asm volatile(".rept 24576\n v_add_u32 %0, 0x12345, %0\n .endr" : "+v"(x));
    const uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (x == 0xdeadbeefu) out[0] = x;
}

// The same straight-line code from THREE wavefronts per workgroup on 105 workgroups (the key-cache QC's shape), on streams
// confined to different sets of CUs: which neighbours share what feeds a wavefront its instructions?  Every wavefront reports
// its time and where it ran (HW_ID: SIMD [5:4], CU [11:8], SH [12], SE [15:13]; XCC_ID).
struct where { uint64_t ticks; uint32_t hw, xcc; };
__global__ void __launch_bounds__(192) k_small3(where *out, uint32_t seed) {
    uint32_t x = seed + threadIdx.x;
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    __syncthreads();
    const uint64_t t0 = wall_clock64();
    for (int p = 0; p < PASSES; p++) asm volatile(".rept 3072\n v_add_u32 %0, 0x12345, %0\n .endr" : "+v"(x));
    const uint64_t t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) { where w; w.ticks = t1 - t0; w.hw = hw; w.xcc = xcc & 15; out[blockIdx.x * 3 + (threadIdx.x >> 6)] = w; }
    if (x == 0xdeadbeefu) out[0].hw = x;
}
static void masked(const char *name, const uint32_t mask[8], int wall_khz) {
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("%-34s stream with CU mask refused\n", name); (void)hipGetLastError(); return; }
    const unsigned wgs = 105;
    where *d; CK(hipMalloc(&d, sizeof(where) * wgs * 3));
    std::vector<where> h(wgs * 3);
    std::vector<double> ns;
    for (int rep = 0; rep < 4; rep++) { hipLaunchKernelGGL(k_small3, dim3(wgs), dim3(192), 0, st, d, 7u); CK(hipStreamSynchronize(st)); }
    CK(hipMemcpy(h.data(), d, sizeof(where) * h.size(), hipMemcpyDeviceToHost));
    std::vector<uint32_t> cus;
    for (const where &w : h) {
        ns.push_back((double)w.ticks / wall_khz * 1e6 / (3072.0 * PASSES));
        cus.push_back((w.xcc << 12) | (((w.hw >> 13) & 7) << 8) | (((w.hw >> 12) & 1) << 4) | ((w.hw >> 8) & 15));
    }
    std::sort(ns.begin(), ns.end());
    std::sort(cus.begin(), cus.end()); cus.erase(std::unique(cus.begin(), cus.end()), cus.end());
    int pairs = 0, xccs[16] = {0};                      // used CUs whose id differs only in the lowest bit (2k, 2k + 1 of one array)
    for (size_t i = 0; i + 1 < cus.size(); i++) if ((cus[i] ^ cus[i + 1]) == 1 && !(cus[i] & 1)) pairs++;
    for (uint32_t c : cus) xccs[c >> 12]++;
    printf("%-34s %3zu CUs used (per XCC %d %d %d %d %d %d %d %d), %2d neighbour pairs (2k, 2k+1): %6.3f ns per instruction median, %6.3f slowest, %6.3f fastest\n",
           name, cus.size(), xccs[0], xccs[1], xccs[2], xccs[3], xccs[4], xccs[5], xccs[6], xccs[7], pairs, ns[ns.size() / 2], ns.back(), ns.front());
    CK(hipFree(d));
    CK(hipStreamDestroy(st));
}

int main(int argc, char **argv) {
    const bool quick = argc > 1 && argv[1][0] == 'q';      // the class marker only: no masked placements
    CK(hipSetDevice(0));
    int wall_khz = 0;
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    uint64_t *d_out; CK(hipMalloc(&d_out, 8 * 1024));
    std::vector<uint64_t> h(1024);
    printf("one wavefront per workgroup, straight-line 8-byte v_add_u32, %d passes; ns per instruction of the slowest / median wavefront\n", PASSES);
    for (int big = 0; big < 2; big++) {
        const double insts = (big ? 24576.0 : 3072.0) * PASSES;
        for (unsigned wgs : {1u, 105u, 256u}) {
            std::vector<double> worst, med;
            for (int rep = 0; rep < 7; rep++) {
                if (big) hipLaunchKernelGGL(k_large, dim3(wgs), dim3(64), 0, 0, d_out, 7u);
                else hipLaunchKernelGGL(k_small, dim3(wgs), dim3(64), 0, 0, d_out, 7u);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h.data(), d_out, 8 * wgs, hipMemcpyDeviceToHost));
                std::vector<uint64_t> v(h.begin(), h.begin() + wgs);
                std::sort(v.begin(), v.end());
                worst.push_back((double)v.back() / wall_khz * 1e6 / insts);
                med.push_back((double)v[wgs / 2] / wall_khz * 1e6 / insts);
            }
            std::sort(worst.begin(), worst.end()); std::sort(med.begin(), med.end());
            printf("%-8s %4u workgroups: %6.3f ns per instruction (slowest wavefront, median of 7 launches; min %.3f max %.3f), median wavefront %6.3f\n",
                   big ? "192 KB" : "24 KB", wgs, worst[3], worst[0], worst[6], med[3]);
        }
    }
    if (quick) return 0;
    printf("105 workgroups x 3 wavefronts of the 24 KB code on CU-masked streams:\n");
    const uint32_t all[8] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
    uint32_t m[8];
    masked("all 256 CUs", all, wall_khz);
    for (int i = 0; i < 8; i++) m[i] = 0x55555555u; masked("even mask bits (128)", m, wall_khz);
    for (int i = 0; i < 8; i++) m[i] = 0xaaaaaaaau; masked("odd mask bits (128)", m, wall_khz);
    for (int i = 0; i < 8; i++) m[i] = 0x33333333u; masked("bits 0,1 of every 4 (128)", m, wall_khz);
    for (int i = 0; i < 8; i++) m[i] = 0x0f0f0f0fu; masked("bits 0-3 of every 8 (128)", m, wall_khz);
    for (int i = 0; i < 8; i++) m[i] = 0x00ff00ffu; masked("bits 0-7 of every 16 (128)", m, wall_khz);
    for (int i = 0; i < 8; i++) m[i] = 0x0000ffffu; masked("bits 0-15 of every 32 (128)", m, wall_khz);
    for (int i = 0; i < 8; i++) m[i] = i < 4 ? ~0u : 0u; masked("mask bits 0-127", m, wall_khz);
    for (int i = 0; i < 8; i++) m[i] = (i & 1) ? 0u : ~0u; masked("even 32-bit words (128)", m, wall_khz);
    return 0;
}
