// tools/ubench_valu.hip - issue-rate microbenchmark for the gfx950 VALU instructions the
// field arithmetic is built from (the guides list no integer-multiply
// rates).  Each lane runs CH independent dependency chains of one instruction; the grid fills
// every SIMD with `waves` wavefronts.  Prints lane-ops/clk/CU at the measured wall time and
// an assumed 2.4 GHz, and cycles per wave64 instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

#define ITERS 4096
#define CH 8

template <int OP>
__global__ void __launch_bounds__(256) bench(uint32_t *out, uint32_t seed) {
    uint32_t a[CH], b[CH];
    uint64_t c[CH];
    double d[CH];
    for (int i = 0; i < CH; i++) { a[i] = seed + threadIdx.x * 7 + i; b[i] = seed * 3 + i; c[i] = a[i]; d[i] = 1.0 + 1e-9 * a[i]; }
    const uint32_t m = seed | 1;
    const unsigned long long mk = __ballot((threadIdx.x + seed) & 1);       // a lane mask in an SGPR pair
    const uint32_t mv = ((threadIdx.x + seed) & 1) ? 0xffffffffu : 0u;      // the same mask as a per-lane word
    if (OP == 14) asm volatile("s_mov_b64 vcc, %0" :: "s"(mk) : "vcc");
    const double dm = 1.0000001, da = 1e-7;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(m) : "vcc");
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(b[i]));
            if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 5) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(m), "v"(seed) : "vcc");
            if (OP == 6) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dm), "v"(da));
            if (OP == 7) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(c[i]));
            if (OP == 8) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(m));
            if (OP == 9) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 10) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 11) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 12) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(c[i]) : "v"(c[(i + 1) % CH]));
            if (OP == 13) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(b[i]));
            if (OP == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : );
            if (OP == 15) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[0]) : "v"(a[i]), "v"(m) : "vcc");   // one dependent chain
            if (OP == 16) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(mk));
            if (OP == 17) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(mv), "v"(m));
            if (OP == 18) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (OP == 19) asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(m));
            if (OP == 20) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(b[i]));
            if (OP == 21) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 22) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 23) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 24) asm volatile("v_mov_b64 %0, %1" : "+v"(c[i]) : "v"(c[(i + 1) % CH]));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < CH; i++) r ^= a[i] ^ b[i] ^ (uint32_t)c[i] ^ (uint32_t)(c[i] >> 32) ^ (uint32_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

typedef void (*kern_t)(uint32_t *, uint32_t);
struct entry { const char *name; kern_t k; int insts; };

int main(int argc, char **argv) {
    int waves_list[] = {1, 2, 4, 8};
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    entry tab[] = {
        {"v_mad_u64_u32", bench<0>, 1}, {"v_mul_lo_u32", bench<1>, 1}, {"v_mul_hi_u32", bench<2>, 1},
        {"v_mad_u32_u24", bench<3>, 1}, {"v_add_u32", bench<4>, 1}, {"v_add_co+v_addc (pair)", bench<5>, 2},
        {"v_fma_f64", bench<6>, 1}, {"v_lshrrev_b64", bench<7>, 1}, {"v_alignbit_b32", bench<8>, 1},
        {"v_and_b32", bench<9>, 1}, {"v_mul_u32_u24", bench<10>, 1}, {"v_mul_hi_u32_u24", bench<11>, 1},
        {"v_lshl_add_u64", bench<12>, 1}, {"v_fma_f32", bench<13>, 1}, {"v_cndmask_b32", bench<14>, 1},
        {"v_mad_u64_u32 (1 dependent chain)", bench<15>, 1},
        {"v_cndmask_b32_e64 (SGPR-pair mask)", bench<16>, 1}, {"v_bfi_b32", bench<17>, 1},
        {"v_mov_b32_dpp quad_perm", bench<18>, 1}, {"v_add_u32_dpp quad_perm", bench<19>, 1},
        {"v_add3_u32", bench<20>, 1}, {"v_lshl_add_u32", bench<21>, 1}, {"v_xor_b32", bench<22>, 1},
        {"v_mov_b32", bench<23>, 1}, {"v_mov_b64", bench<24>, 1},
    };
    uint32_t *out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-36s %6s %10s %14s %16s\n", "instruction", "w/SIMD", "ms", "laneops/clk/CU", "cyc/wave-inst/SIMD");
    for (auto &t : tab) for (int w : waves_list) {
        // 256-thread blocks = 4 waves = one per SIMD; w blocks per CU
        dim3 grid(cus * w), block(256);
        hipLaunchKernelGGL(t.k, grid, block, 0, 0, out, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(t.k, grid, block, 0, 0, out, 12345u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ops = (double)cus * w * 256 * ITERS * CH;          // lane-ops (asm statements)
        const double clk = ms * 1e-3 * 2.4e9;
        const double per_cu = ops / clk / cus;
        const double cyc_per_inst = clk / ((double)w * ITERS * CH * t.insts);   // per SIMD: w waves each ITERS*CH*insts instructions
        printf("%-36s %6d %10.3f %14.2f %16.2f\n", t.name, w, ms, per_cu, cyc_per_inst);
    }
    return 0;
}
