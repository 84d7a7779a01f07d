// tools/mem_latency_probe.hip - how long does ONE wavefront wait for a dependent gather on this box?
//
// A QC's latency is one wavefront's chain, and on that chain sit dependent table lookups that nothing can
// cover: 24 entries of the 5.3 GB key table per lane of the key-cache kernel, 8 of the base point's 67 MB table per quad of
// the cold kernel - plus the instruction fetches of ~100 K straight-line instructions.  Round 4 found two boxes with the SAME
// shader clock for a lone wavefront (tools/clock_probe: 2,396 MHz, 10.0 cycles per dependent multiply-add on both) whose
// key-cache QC p50 nevertheless differed by 46 us (0.087 against 0.134 ms) while every throughput figure agreed: whatever
// distinguishes the "classes of box" is not the clock and only hurts a LONE chain.  This probe measures the memory side of
// such a chain: each of the 64 lanes of one wavefront walks its own random cycle through a buffer, one 16-byte load per hop,
// every hop dependent on the previous one - over buffers of 32 MB (fits the L2s' reach / MALL), 256 MB, 2 GB and 6 GB (the key
// table's size: HBM + TLB reach), allocated in one piece.  Reported: ns per hop (wall-clock counter) and shader cycles per hop.
//   hipcc --offload-arch=gfx950 -O2 tools/mem_latency_probe.hip -o tools/mem_latency_probe && tools/mem_latency_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// next[i] = index of the 128-byte line that follows line i (first dword of the line); lines is a power of two.
// The walk x -> (a x + c) mod lines with a = 1 mod 4, c odd is a full cycle (Hull-Dobell): no table of successors to build
// on the host, every line visited once per period, successive lines far apart.
__global__ void k_fill(uint32_t *buf, uint64_t lines) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < lines; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t nxt = (i * 0x9e3779b1ull * 4 + i + 0x7f4a7c15ull * 2 + 1) & (lines - 1);      // a = 4 k + 1, c odd
        uint4 *p = reinterpret_cast<uint4 *>(buf + i * 32);
        p[0] = make_uint4((uint32_t)nxt, (uint32_t)(nxt >> 32), 0, 0);
    }
}
__global__ void __launch_bounds__(64) k_chase(const uint32_t *__restrict__ buf, uint64_t lines, uint32_t hops, uint64_t *out) {
    uint64_t x = ((uint64_t)threadIdx.x * 0x9e3779b97f4a7c15ull) & (lines - 1);       // 64 different starting points
    const uint64_t w0 = wall_clock64(), t0 = clock64();
    for (uint32_t h = 0; h < hops; h++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(buf + x * 32);                // one 16-byte load per hop, as a table entry's first piece
        x = ((uint64_t)v.y << 32 | v.x) & (lines - 1);
    }
    const uint64_t t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
    if (x == ~0ull) out[2] = x;
}

// The throughput side: `G` wavefronts, every lane fetching INDEPENDENT random 128-byte lines (eight 16-byte loads in flight
// per lane and step, as a key-table entry is read) - what the key-cache kernel's 6,667 x 48 lookups look like to the memory
// system when the chip is full.  Lines per microsecond over the same footprints.
__global__ void __launch_bounds__(64) k_gather(const uint32_t *__restrict__ buf, uint64_t lines, uint32_t steps, uint32_t *sink) {
    uint64_t x = ((uint64_t)(blockIdx.x * 64 + threadIdx.x) * 0x9e3779b97f4a7c15ull) >> 7;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; s++) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            v[j] = *reinterpret_cast<const uint4 *>(buf + ((x >> 20) & (lines - 1)) * 32 + 4 * (j & 7));
        }
#pragma unroll
        for (int j = 0; j < 8; j++) acc += v[j].x ^ v[j].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv) {
    const bool quick = argc > 1 && argv[1][0] == 'q';      // three sizes, no throughput runs (bench.py's env record)
    CK(hipSetDevice(0));
    int wall_khz = 0;
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    uint64_t *d_out, h_out[3];
    CK(hipMalloc(&d_out, 64));
    const uint32_t hops = 2000;
    printf("one wavefront, 64 lanes each on its own walk, %u dependent 16-byte loads per lane (one per 128-byte line)\n", hops);
    const size_t sizes_mb[] = {2, 32, 256, 2048, 6144, 12288};
    for (size_t mb : sizes_mb) {
        if (quick && mb != 2 && mb != 256 && mb != 6144) continue;
        uint64_t lines = (mb << 20) / 128;
        uint64_t p2 = 1; while (p2 * 2 <= lines) p2 *= 2;        // the walk needs a power of two; allocate what is asked, walk the power of two inside
        uint32_t *buf = nullptr;
        if (hipMalloc(&buf, mb << 20) != hipSuccess) { printf("%6zu MB: allocation failed\n", mb); (void)hipGetLastError(); continue; }
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, buf, p2);
        CK(hipDeviceSynchronize());
        std::vector<double> ns, cyc;
        for (int rep = 0; rep < 7; rep++) {
            hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, buf, p2, hops, d_out);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h_out, d_out, 24, hipMemcpyDeviceToHost));
            ns.push_back((double)h_out[1] / wall_khz * 1e6 / hops);
            cyc.push_back((double)h_out[0] / hops);
        }
        std::sort(ns.begin(), ns.end()); std::sort(cyc.begin(), cyc.end());
        printf("%6zu MB (walk over %5.0f MB): %7.1f ns per hop (min %.1f max %.1f), %7.1f shader cycles per hop\n", mb, (double)p2 * 128 / (1 << 20),
               ns[3], ns[0], ns[6], cyc[3]);
        // independent gathers: one wavefront per SIMD (1,024) and four (4,096), 64 x 8 lines per lane
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (unsigned G : {1024u, 4096u}) {
            if (quick) break;
            const uint32_t steps = 64;
            std::vector<double> us;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k_gather, dim3(G), dim3(64), 0, 0, buf, p2, steps, (uint32_t *)d_out);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); us.push_back(ms * 1e3);
            }
            std::sort(us.begin(), us.end());
            const double nlines = (double)G * 64 * steps * 8;
            printf("          independent gathers, %4u wavefronts: %8.1f lines per us (%6.1f GB/s of 128-byte lines), %.1f us per launch\n", G, nlines / us[2],
                   nlines * 128 / us[2] / 1e3, us[2]);
        }
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        fflush(stdout);
        CK(hipFree(buf));
    }
    return 0;
}
