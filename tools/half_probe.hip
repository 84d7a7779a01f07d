// half_probe.hip - the half-size scalar reduction (halfscalar.h) on the GPU against the same header compiled for the host:
// identical (u, v) for every k, and the histogram of window counts the cold kernels will run.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I hotstuff_amd/csrc tools/half_probe.hip -o tools/half_probe && tools/half_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "halfscalar.h"
using namespace hs;

__host__ __device__ static void make_k(u32 k[8], u32 i) {
    u32 in[24], h[16];
    for (int j = 0; j < 24; j++) in[j] = 0x9e3779b9u * (i + 1) + j * 0x85ebca6bu;
    sha512_96(h, in);
    sc_reduce512(k, h);
}
struct res { u32 u[8], v[8], u_neg, nwin, k[8]; };
__host__ __device__ static void one(res &r, u32 i) {
    u32 k[8]; make_k(k, i);
    half_pair h; half_scalar(h, k);
    u32 ru[8], rv[8];
    sc_recode16(ru, h.u); sc_recode16(rv, h.v);
    const int a = sc_recode16_len(ru), b = sc_recode16_len(rv);
    for (int j = 0; j < 8; j++) { r.u[j] = h.u[j]; r.v[j] = h.v[j]; }
    r.u_neg = h.u_neg; r.nwin = (u32)(a > b ? a : b);
    for (int j = 0; j < 8; j++) r.k[j] = k[j];
}
__global__ void k_probe(res *out, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) one(out[i], i);
}
// timing of the scalar stage on lone wavefronts (one per SIMD): hash only, hash + reduction, hash + reduction + v s mod L + recodings
template <int WHAT>
__global__ void k_time(u32 *sink, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    u32 k[8]; make_k(k, i);
    u32 acc = k[0];
    if (WHAT >= 1) { half_pair h; half_scalar(h, k); acc ^= h.u[0] ^ h.v[0] ^ h.u_neg;
        if (WHAT >= 2) { u32 w[8], ru[8], rv[8], rw[8]; sc_mul(w, h.v, k); sc_recode16(ru, h.u); sc_recode16(rv, h.v); sc_recode65536(rw, w);
            acc ^= rw[3] ^ (u32)sc_recode16_len(ru) ^ (u32)sc_recode16_len(rv); } }
    if (acc == 0x12345u) sink[0] = acc;
}
template <int WHAT>
static float time_it(u32 *sink, u32 waves) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_time<WHAT>, dim3(waves), dim3(64), 0, 0, sink, waves * 64);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k_time<WHAT>, dim3(waves), dim3(64), 0, 0, sink, waves * 64);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / 10 * 1000.0f;
}
int main() {
    {
        u32 *sink = nullptr; hipMalloc((void **)&sink, 64);
        for (u32 waves : {512u, 1024u, 2048u, 8192u})
            printf("scalar stage, %u wavefronts: hash %.1f us, + half_scalar %.1f us, + v s mod L and recodings %.1f us\n", waves,
                   time_it<0>(sink, waves), time_it<1>(sink, waves), time_it<2>(sink, waves));
    }
    const u32 n = 1 << 16;
    res *d = nullptr;
    if (hipMalloc((void **)&d, n * sizeof(res)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_probe, dim3(n / 64), dim3(64), 0, 0, d, n);
    std::vector<res> g(n);
    if (hipMemcpy(g.data(), d, n * sizeof(res), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
    u32 diff = 0, hist[65] = {0}, whist[65] = {0};
    for (u32 i = 0; i < n; i++) {
        res h; one(h, i);
        if (memcmp(&h, &g[i], sizeof(res))) {
            if (diff < 3) {
                printf("MISMATCH at %u: gpu nwin %u host nwin %u\n  gpu  u", i, g[i].nwin, h.nwin);
                for (int j = 7; j >= 0; j--) printf(" %08x", g[i].u[j]);
                printf(" neg %u k0 %08x v", g[i].u_neg, g[i].k[0]);
                for (int j = 7; j >= 0; j--) printf(" %08x", g[i].v[j]);
                printf("\n  host u");
                for (int j = 7; j >= 0; j--) printf(" %08x", h.u[j]);
                printf(" neg %u k0 %08x v", h.u_neg, h.k[0]);
                for (int j = 7; j >= 0; j--) printf(" %08x", h.v[j]);
                printf("\n");
            }
            diff++;
        }
        hist[g[i].nwin > 64 ? 64 : g[i].nwin]++;
    }
    for (u32 w = 0; w < n / 64; w++) { u32 m = 0; for (u32 l = 0; l < 64; l++) m = g[w * 64 + l].nwin > m ? g[w * 64 + l].nwin : m; whist[m > 64 ? 64 : m]++; }
    printf("mismatches %u of %u\nper lane:", diff, n);
    for (int i = 0; i <= 64; i++) if (hist[i]) printf(" %d:%u", i, hist[i]);
    printf("\nper wavefront of 64:");
    for (int i = 0; i <= 64; i++) if (whist[i]) printf(" %d:%u", i, whist[i]);
    printf("\n");
    return diff ? 2 : 0;
}
