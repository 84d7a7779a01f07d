// ubench_gather.hip - calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in THIS project's access patterns
// ("calibrate on a known copy first"; MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide
// coalesced read, other widths and WRITE_SIZE are uncalibrated).  Every kernel moves a KNOWN volume between HBM and
// the chip - buffers of 4 GiB, far beyond the 32 MiB of L2 and the 256 MiB Infinity Cache, each byte touched once:
//
//   k_cal_stream_read16   wide coalesced read, 16 B per lane (the guide's reference case)
//   k_cal_stream_read4    coalesced dword rows (the SoA workspace rows of k_verify: 64 consecutive dwords per wavefront)
//   k_cal_gather_line16   every lane reads ONE random 128-byte line with eight 16-byte loads (a table entry of the
//                         key-cache kernels, a Niels record of the RLC accumulate stage)
//   k_cal_gather_entry160 every lane reads ten 16-byte pieces of a random 160-byte entry (the one-lane kernels' table)
//   k_cal_gather_dword    every lane reads one random dword (its line is fetched for 4 useful bytes)
//   k_cal_stream_write16  wide coalesced write, 16 B per lane
//   k_cal_row_write4      coalesced dword rows (k_prepare's SoA stores)
//   k_cal_entry_write16   every lane writes ten 16-byte pieces to its own 160-byte entry, lanes 1,440 B apart
//                         (ws_store_cached: the per-signature table of the one-lane kernels)
//
// Prints one JSON object with each kernel's useful bytes and touched-line bytes; run it under
//   rocprofv3 --pmc FETCH_SIZE -d out -o cal -- tools/ubench_gather      (and a second pass with WRITE_SIZE)
// and divide (tools/pmc_calibrate.py).   Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_gather.hip -o tools/ubench_gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {      // splitmix64: a fixed pseudo-random permutation-like map
    x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31);
}
__global__ void k_cal_stream_read16(const uint4 *__restrict__ p, size_t n16, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_cal_stream_read4(const uint32_t *__restrict__ p, size_t n4, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) *sink = acc;
}
// lines: number of 128-byte lines in the buffer (a power of two); every lane visits `per_lane` distinct random lines.
// The map lane -> line is a bijection on [0, lines) (multiplication by an odd constant mod 2^k), so no line is read twice.
__global__ void k_cal_gather_line16(const uint4 *__restrict__ p, uint64_t lines, uint32_t per_lane, uint32_t *sink) {
    uint32_t acc = 0;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, total = (uint64_t)gridDim.x * blockDim.x;
    for (uint32_t r = 0; r < per_lane; r++) {
        const uint64_t line = ((t + (uint64_t)r * total) * 0x9e3779b97f4a7c15ull + 12345) & (lines - 1);
        const uint4 *q = p + line * 8;
#pragma unroll
        for (int j = 0; j < 8; j++) { const uint4 v = q[j]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) *sink = acc;
}
// entries of 160 bytes back to back (they straddle 128-byte lines as the one-lane kernels' table does)
__global__ void k_cal_gather_entry160(const uint4 *__restrict__ p, uint64_t entries_pow2, uint32_t per_lane, uint32_t *sink) {
    uint32_t acc = 0;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, total = (uint64_t)gridDim.x * blockDim.x;
    for (uint32_t r = 0; r < per_lane; r++) {
        const uint64_t e = ((t + (uint64_t)r * total) * 0x9e3779b97f4a7c15ull + 777) & (entries_pow2 - 1);
        const uint4 *q = p + e * 10;
#pragma unroll
        for (int j = 0; j < 10; j++) { const uint4 v = q[j]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_cal_gather_dword(const uint32_t *__restrict__ p, uint64_t lines, uint32_t per_lane, uint32_t *sink) {
    uint32_t acc = 0;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, total = (uint64_t)gridDim.x * blockDim.x;
    for (uint32_t r = 0; r < per_lane; r++) {
        const uint64_t line = ((t + (uint64_t)r * total) * 0x9e3779b97f4a7c15ull + 999) & (lines - 1);
        acc ^= p[line * 32 + (mix(line) & 31)];
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_cal_stream_write16(uint4 *__restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ void k_cal_row_write4(uint32_t *__restrict__ p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
// lane t owns the 1,440-byte slice t; it writes `entries` (<= 9) 160-byte entries of it, ten 16-byte stores each
__global__ void k_cal_entry_write16(uint4 *__restrict__ p, uint64_t lanes, uint32_t entries) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lanes) return;
    uint4 *q = p + t * 90;
    for (uint32_t e = 0; e < entries; e++)
#pragma unroll
        for (int j = 0; j < 10; j++) q[e * 10 + j] = make_uint4((uint32_t)t, e, j, 7);
}

int main(int argc, char **argv) {
    const size_t GiB = (size_t)1 << 30, bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 4) * GiB;
    uint8_t *buf = nullptr; uint32_t *sink = nullptr;
    CHECK(hipMalloc((void **)&buf, bytes));
    CHECK(hipMalloc((void **)&sink, 4));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    const uint64_t lines = bytes / 128;              // a power of two for 1, 2, 4, 8 GiB
    uint64_t ent_pow2 = 1; while (ent_pow2 * 2 * 160 <= bytes) ent_pow2 *= 2;
    const int grid = 256 * 32, block = 256;
    const uint64_t lanes = (uint64_t)grid * block;   // 2,097,152
    const uint32_t per_line = (uint32_t)(lines / lanes / 2);         // half the lines: 2 GiB of a 4 GiB buffer
    const uint32_t per_ent = (uint32_t)(ent_pow2 / lanes / 2);
    const uint32_t per_dw = (uint32_t)(lines / lanes / 2);
    const uint64_t wlanes = bytes / 1440;
    // every kernel twice: rocprofv3 averages per dispatch, the first dispatch also pays cold TLBs
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_cal_stream_read16, dim3(grid), dim3(block), 0, 0, (const uint4 *)buf, bytes / 16, sink);
        hipLaunchKernelGGL(k_cal_stream_read4, dim3(grid), dim3(block), 0, 0, (const uint32_t *)buf, bytes / 4, sink);
        hipLaunchKernelGGL(k_cal_gather_line16, dim3(grid), dim3(block), 0, 0, (const uint4 *)buf, lines, per_line, sink);
        hipLaunchKernelGGL(k_cal_gather_entry160, dim3(grid), dim3(block), 0, 0, (const uint4 *)buf, ent_pow2, per_ent, sink);
        hipLaunchKernelGGL(k_cal_gather_dword, dim3(grid), dim3(block), 0, 0, (const uint32_t *)buf, lines, per_dw, sink);
        hipLaunchKernelGGL(k_cal_stream_write16, dim3(grid), dim3(block), 0, 0, (uint4 *)buf, bytes / 16);
        hipLaunchKernelGGL(k_cal_row_write4, dim3(grid), dim3(block), 0, 0, (uint32_t *)buf, bytes / 4);
        hipLaunchKernelGGL(k_cal_entry_write16, dim3((unsigned)((wlanes + 255) / 256)), dim3(256), 0, 0, (uint4 *)buf, wlanes, 9u);
        CHECK(hipDeviceSynchronize());
    }
    printf("{\"buffer_bytes\": %zu, \"kernels\": {"
           "\"k_cal_stream_read16\": {\"read_bytes\": %zu}, "
           "\"k_cal_stream_read4\": {\"read_bytes\": %zu}, "
           "\"k_cal_gather_line16\": {\"read_bytes\": %llu, \"note\": \"one whole 128-byte line per gather, eight 16-byte loads\"}, "
           "\"k_cal_gather_entry160\": {\"read_bytes\": %llu, \"touched_line_bytes_128\": %llu, \"touched_line_bytes_64\": %llu, \"note\": \"160-byte entries back to back at multiples of 32 B: two 128-byte lines / three 64-byte sectors each\"}, "
           "\"k_cal_gather_dword\": {\"read_bytes\": %llu, \"touched_line_bytes_128\": %llu, \"touched_line_bytes_64\": %llu}, "
           "\"k_cal_stream_write16\": {\"write_bytes\": %zu}, "
           "\"k_cal_row_write4\": {\"write_bytes\": %zu}, "
           "\"k_cal_entry_write16\": {\"write_bytes\": %llu}}}\n",
           bytes, bytes, bytes,
           (unsigned long long)lanes * per_line * 128,
           (unsigned long long)lanes * per_ent * 160, (unsigned long long)lanes * per_ent * 256, (unsigned long long)lanes * per_ent * 192,
           (unsigned long long)lanes * per_dw * 4, (unsigned long long)lanes * per_dw * 128, (unsigned long long)lanes * per_dw * 64,
           bytes, bytes, (unsigned long long)wlanes * 1440);
    CHECK(hipFree(buf));
    return 0;
}
