// tools/dpp_subrev_check.hip - what gfx950 does with DPP on the "rev" subtract.
//
// LLVM's DPP combiner folds   x - mov_dpp(y)   into   v_subrev_u32_dpp d, y, x  (D = S1 - S0, DPP on S0),
// assuming the lane permutation applies to S0 = y.  On MI355X the permutation lands on the OTHER
// operand: D[lane] = x[perm(lane)] - y[lane].  v_sub_u32_dpp, v_add_u32_dpp and v_xor_b32_dpp behave as
// documented.  Found when a rewritten doubling accepted every signature (Z = 0); the engine avoids
// the pattern (verify_quad.h q_comb) and the build refuses code objects that contain it.
//   hipcc --offload-arch=gfx950 -O2 tools/dpp_subrev_check.hip -o dpp_check && ./dpp_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t *out) {
    uint32_t a = 100 * (threadIdx.x + 1), b = 7 * (threadIdx.x + 1) + 100000, r[6];
    asm volatile("s_nop 4\n\tv_sub_u32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4" : "=v"(r[0]) : "v"(a), "v"(b));
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4" : "=v"(r[1]) : "v"(a), "v"(b));
    asm volatile("s_nop 4\n\tv_add_u32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4" : "=v"(r[2]) : "v"(a), "v"(b));
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 quad_perm:[0,1,1,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4" : "=v"(r[3]) : "v"(a), "v"(b));
    // in-place form the compiler emitted: vdst == src1
    uint32_t c = b;
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %0 quad_perm:[0,1,1,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4" : "+v"(c) : "v"(a));
    r[4] = c;
    asm volatile("s_nop 4\n\tv_xor_b32_dpp %0, %1, %2 quad_perm:[0,1,1,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4" : "=v"(r[5]) : "v"(a), "v"(b));
    for (int i = 0; i < 6; i++) out[i * 64 + threadIdx.x] = r[i];
}
int main() {
    uint32_t *d; hipMalloc(&d, 6 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[6 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *nm[6] = {"sub_dpp[1,0,3,2]   exp dpp(a)-b", "subrev_dpp[1,0,3,2] exp b-dpp(a)", "add_dpp[1,0,3,2]   exp dpp(a)+b", "subrev_dpp[0,1,1,0] exp b-dpp(a)", "subrev inplace [0,1,1,0]", "xor_dpp[0,1,1,0]"};
    for (int l = 0; l < 4; l++) printf("lane %d: a=%u b=%u\n", l, 100 * (l + 1), 7 * (l + 1) + 100000);
    for (int i = 0; i < 6; i++) { printf("%-36s:", nm[i]); for (int l = 0; l < 4; l++) printf(" %d", (int)h[i * 64 + l]); printf("\n"); }
    return 0;
}
