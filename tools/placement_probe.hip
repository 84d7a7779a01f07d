// tools/placement_probe.hip - WHERE do the wavefronts of a QC-sized grid run?
//
// A QC's latency is the chain of a lone wavefront, and a SIMD shared by three wavefronts gives each a third of its issue slots
// where two still get a lone wavefront's cadence (profiles/r1_ubench_valu.txt: 11.4 cycles per multiply-add alone, 5.9 per SIMD
// with two).  The key-cache kernel's grid for a 10,000-validator QC is 105 workgroups of three wavefronts - far fewer than the
// chip's 1,024 SIMDs - so whether its chains run alone depends entirely on where the dispatcher puts them, which HIP promises
// nothing about.  This probe launches grids of the shapes the latency kernels use, keeps every wavefront alive long enough for
// the whole grid to be resident at once, and has each wavefront report its XCC / SE / CU / SIMD (s_getreg HW_ID, XCC_ID).
// Printed per shape: CUs used, workgroups per CU, wavefronts per SIMD, and how many workgroups have two of their wavefronts on
// one SIMD - the facts that can differ between two boxes whose clocks agree.
//   hipcc --offload-arch=gfx950 -O2 tools/placement_probe.hip -o tools/placement_probe && tools/placement_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct rec { uint32_t hw_id, xcc, wg, wave; uint64_t t0, t1; };

template <int LDS_WORDS>
__global__ void k_where(rec *out, uint32_t spin_ticks) {
    __shared__ uint32_t s_pad[LDS_WORDS];
    const uint64_t t0 = wall_clock64();
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    s_pad[threadIdx.x % LDS_WORDS] = hw;
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        rec r; r.hw_id = hw; r.xcc = xcc & 0xf; r.wg = blockIdx.x; r.wave = threadIdx.x >> 6; r.t0 = t0; r.t1 = wall_clock64();
        out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r;
    }
    if (s_pad[0] == 0xdeadbeefu) out[0].hw_id = 0;
}

static void report(const char *name, const std::vector<rec> &v, int waves_per_wg, int wall_khz) {
    // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
    std::map<uint32_t, int> per_simd, per_cu;
    std::map<uint32_t, std::vector<uint32_t>> wg_simds;
    std::map<uint32_t, int> per_xcc;
    uint64_t tmin = ~0ull, tstart_max = 0;
    for (const rec &r : v) {
        const uint32_t simd = (r.hw_id >> 4) & 3, cu = (r.hw_id >> 8) & 15, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7;
        const uint32_t cukey = (r.xcc << 12) | (se << 8) | (sh << 4) | cu;
        per_cu[cukey]++; per_simd[(cukey << 2) | simd]++; per_xcc[r.xcc]++;
        wg_simds[r.wg].push_back((cukey << 2) | simd);
        tmin = std::min(tmin, r.t0); tstart_max = std::max(tstart_max, r.t0);
    }
    int hist_simd[9] = {0}, hist_cu_wg[9] = {0}, shared_wgs = 0;
    for (auto &p : per_simd) hist_simd[std::min(p.second, 8)]++;
    for (auto &p : per_cu) hist_cu_wg[std::min(p.second / waves_per_wg, 8)]++;
    for (auto &p : wg_simds) { std::sort(p.second.begin(), p.second.end()); if (std::adjacent_find(p.second.begin(), p.second.end()) != p.second.end()) shared_wgs++; }
    printf("%-46s %4zu wavefronts: %3zu CUs used (workgroups per CU: 1:%d 2:%d 3:%d 4+:%d); SIMDs holding 1 / 2 / 3 / 4+ wavefronts: %d / %d / %d / %d; "
           "workgroups with two wavefronts on one SIMD: %d; XCCs used %zu; last start %.1f us after the first\n",
           name, v.size(), per_cu.size(), hist_cu_wg[1], hist_cu_wg[2], hist_cu_wg[3], hist_cu_wg[4] + hist_cu_wg[5] + hist_cu_wg[6] + hist_cu_wg[7] + hist_cu_wg[8],
           hist_simd[1], hist_simd[2], hist_simd[3], hist_simd[4] + hist_simd[5] + hist_simd[6] + hist_simd[7] + hist_simd[8], shared_wgs, per_xcc.size(),
           (double)(tstart_max - tmin) / wall_khz * 1e3);
}

// Two-wavefront workgroups (the split eight-lane cold kernel: wavefront 0 = the MAIN chain, ~270 us; wavefront 1 = the helper,
// ~50 us): after the helpers have gone, how many SIMDs hold two MAIN wavefronts while a neighbour SIMD of the same CU idles -
// under the kernel's rule (main = wavefront 0) and under other rules a workgroup could apply from what it can see (its index,
// its wavefronts' SIMD ids); "best" = a per-CU greedy assignment with full knowledge, the bound.
static void report_roles(const char *name, const std::vector<rec> &v) {
    struct wgp { uint32_t cu, simd[2]; };
    std::map<uint32_t, wgp> wg;
    for (const rec &r : v) {
        const uint32_t simd = (r.hw_id >> 4) & 3, cu = (r.hw_id >> 8) & 15, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7;
        wgp &w = wg[r.wg]; w.cu = (r.xcc << 12) | (se << 8) | (sh << 4) | cu; w.simd[r.wave & 1] = simd;
    }
    const char *rules[] = { "main = wavefront 0 (the kernel)", "wavefront (wg & 1)", "wavefront ((wg >> 8) & 1)", "wavefront ((wg >> 9) & 1)",
                            "the wavefront on the odd SIMD when (wg >> 8) is odd", "the wavefront on SIMD (wg >> 8) & 3 if it has one, else wavefront 0" };
    for (int rule = 0; rule < 6; rule++) {
        std::map<uint32_t, int> mains;
        for (auto &p : wg) {
            const uint32_t i = p.first; const wgp &w = p.second;
            int m = 0;
            if (rule == 1) m = i & 1; else if (rule == 2) m = (i >> 8) & 1; else if (rule == 3) m = (i >> 9) & 1;
            else if (rule == 4) { const uint32_t want = (i >> 8) & 1; m = (w.simd[0] & 1) == want ? 0 : ((w.simd[1] & 1) == want ? 1 : 0); }
            else if (rule == 5) { const uint32_t want = (i >> 8) & 3; m = w.simd[0] == want ? 0 : (w.simd[1] == want ? 1 : 0); }
            mains[(w.cu << 2) | w.simd[m]]++;
        }
        int h[5] = {0};
        for (auto &q : mains) h[std::min(q.second, 4)]++;
        printf("    %-72s SIMDs holding 1 / 2 / 3 / 4+ main wavefronts: %d / %d / %d / %d\n", rules[rule], h[1], h[2], h[3], h[4]);
    }
    // the bound: per CU, workgroups in index order, each takes whichever of its two SIMDs holds fewer mains so far
    std::map<uint32_t, int> mains;
    for (auto &p : wg) { const wgp &w = p.second; const uint32_t a = (w.cu << 2) | w.simd[0], b = (w.cu << 2) | w.simd[1]; mains[mains[a] <= mains[b] ? a : b]++; }
    int h[5] = {0};
    for (auto &q : mains) if (q.second) h[std::min(q.second, 4)]++;
    printf("    %-72s SIMDs holding 1 / 2 / 3 / 4+ main wavefronts: %d / %d / %d / %d\n", "greedy with full knowledge (bound)", h[1], h[2], h[3], h[4]);
    // one CU's workgroups in index order: (workgroup: SIMD of wavefront 0, SIMD of wavefront 1)
    const uint32_t cu0 = wg.begin()->second.cu;
    printf("    the workgroups on one CU:");
    for (auto &p : wg) if (p.second.cu == cu0) printf("  %u:(%u,%u)", p.first, p.second.simd[0], p.second.simd[1]);
    printf("\n");
    (void)name;
}

template <int LDS_WORDS>
static void run(const char *name, unsigned wgs, unsigned threads, int wall_khz, hipStream_t st) {
    const unsigned wpw = threads / 64;
    rec *d; CK(hipMalloc(&d, sizeof(rec) * wgs * wpw));
    std::vector<rec> h(wgs * wpw);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_where<LDS_WORDS>, dim3(wgs), dim3(threads), 0, st, d, (uint32_t)((long long)wall_khz * 60 / 1000));   // 60 us
        CK(hipStreamSynchronize(st));
    }
    CK(hipMemcpy(h.data(), d, sizeof(rec) * h.size(), hipMemcpyDeviceToHost));
    report(name, h, (int)wpw, wall_khz);
    if (wpw == 2) report_roles(name, h);
    CK(hipFree(d));
}

int main() {
    CK(hipSetDevice(0));
    int wall_khz = 0;
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    hipStream_t st; CK(hipStreamCreate(&st));
    printf("wall clock %d kHz; every wavefront stays ~60 us so the whole grid is resident at once\n", wall_khz);
    // the key-cache latency kernel at N = 10,000: 105 workgroups x 3 wavefronts, 5.3 KB of LDS
    run<1360>("key cache QC N=10,000 (105 x 192, 5.3 KB LDS)", 105, 192, wall_khz, st);
    run<1360>("key cache QC N=100,000 (1042 x 192)", 1042, 192, wall_khz, st);
    // the eight-lane cold kernel: 6,667 signatures = 834 wavefronts of 8, 20 KB of LDS each; split form: 2 wavefronts per workgroup
    run<5120>("cold QC N=10,000, one wavefront (834 x 64, 20 KB)", 834, 64, wall_khz, st);
    run<5120>("cold QC N=10,000, split (834 x 128, 20 KB)", 834, 128, wall_khz, st);
    // a 1,024-signature batch in the quad form: 64 wavefronts
    run<5120>("1,024-signature batch, quad form (64 x 64)", 64, 64, wall_khz, st);
    run<64>("256 x 256 (one workgroup per CU?)", 256, 256, wall_khz, st);
    return 0;
}
