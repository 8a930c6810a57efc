// Secondary roofline of the verify path ("Fp-mulmods/s achieved / peak, with peak measured by a register-resident
// Montgomery-mul microbenchmark on gfx950 -- do not assume an ISA rate").  The kernels below run nothing but the library's own
// multiplication primitives on register-resident operands, two independent chains per lane:
//   kind 0  fp_mul        one 81-term column product + one Montgomery reduction per call                (1 mulmod per lane-call)
//   kind 1  f2_mul_lane   the lane-pair Fp2 product: two column products + one reduction per lane       (2 mulmods per lane-call,
//                         the unit the op counter of tests/host_sim counts for the pair kernels)
//   kind 2 / 3 / 4        the other primitives of the instruction stream, counted per lane-call (kind 1's factor 2 does not apply):
//                         modular add / sub as single carry chains, as interleaved pairs (fp_add_x2 / fp_sub_x2), and f2_mul_xi
// Rates are measured in TIME (HIP events around the launch), so no clock frequency is assumed anywhere; the shader clock under
// this load is reported separately from s_memtime / s_memrealtime when the two counters differ.
#define ZKV_PAIRED 1
#undef ZKV_FP_MUL_NOINLINE
#include "zkv_internal.h"

namespace zkv {

template <int KIND>
__global__ __launch_bounds__(64) void k_diag_mulmod(uint32_t iters, uint32_t* __restrict__ out, unsigned long long* __restrict__ clk) {
    Fp a0, b0, a1, b1;
    for (int i = 0; i < 8; i++) {
        a0.v[i] = 0x9E3779B9u * (threadIdx.x + 1) + i; b0.v[i] = 0x85EBCA6Bu * (blockIdx.x + 3) + 7 * i;
        a1.v[i] = a0.v[i] ^ 0x5A5A5A5Au; b1.v[i] = b0.v[i] + 0x01010101u;
    }
    a0.v[7] &= 0x1fffffffu; b0.v[7] &= 0x1fffffffu; a1.v[7] &= 0x1fffffffu; b1.v[7] &= 0x1fffffffu;      // < 2^253: inside the multipliers' input range
    unsigned long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll 1
    for (uint32_t i = 0; i < iters; i++) {
        if (KIND == 0) { a0 = fp_mul(a0, b0); a1 = fp_mul(a1, b1); b0 = fp_mul(b0, a1); b1 = fp_mul(b1, a0); }
        else if (KIND == 1) { a0 = f2_mul_lane(a0, b0); a1 = f2_mul_lane(a1, b1); b0 = f2_mul_lane(b0, a1); b1 = f2_mul_lane(b1, a0); }
        else if (KIND == 2) { a0 = fp_add(a0, b0); a1 = fp_sub(a1, b1); b0 = fp_add(b0, a1); b1 = fp_sub(b1, a0); }          // four single chains
        else if (KIND == 3) {                                                                                             // four ops as two interleaved pairs
            Fp r0, r1; fp_add_x2(a0, b0, a1, b1, r0, r1); a0 = r0; a1 = r1;
            fp_sub_x2(b0, a1, b1, a0, r0, r1); b0 = r0; b1 = r1;
        } else {                                                                                                          // four xi-multiplications
            Fp2 x; x.h = a0; a0 = f2_mul_xi(x).h; x.h = a1; a1 = f2_mul_xi(x).h; x.h = b0; b0 = f2_mul_xi(x).h; x.h = b1; b1 = f2_mul_xi(x).h;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_readcyclecounter() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; i++) r ^= a0.v[i] ^ b0.v[i] ^ a1.v[i] ^ b1.v[i];
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = r;
}

// Independent issue-rate roof (round 3): nothing but v_mad_u64_u32 (kind 0) / v_mad_i64_i32 (kind 1) / v_add_u32 (kind 2: a plain VOP2
// instruction, for scale) on register-resident operands, eight independent accumulator chains per lane, 64 instructions per loop
// trip.  The verify kernels' executed multiply-adds per second divided by the kind-0 rate is the fraction of the multiplier-issue
// bound they reach; unlike roofline.mulmod this roof does not move when the library's own multiplier gets faster.
template <int KIND>
__global__ __launch_bounds__(64) void k_diag_issue(uint32_t iters, uint32_t* __restrict__ out, unsigned long long* __restrict__ clk) {
    uint32_t a[8], b[8]; uint64_t acc[8];
    for (int u = 0; u < 8; u++) { a[u] = 0x9E3779B9u * (threadIdx.x + 1) + u; b[u] = 0x85EBCA6Bu * (blockIdx.x + 3) + 7 * u; acc[u] = a[u]; }
    unsigned long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll 1
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (KIND == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[u]) : "v"(a[u]), "v"(b[(u + r) & 7]) : "vcc");
                else if (KIND == 1) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[u]) : "v"(a[u]), "v"(b[(u + r) & 7]) : "vcc");
                else asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[u]) : "v"(b[(u + r) & 7]));
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_readcyclecounter() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    uint32_t r = 0;
    for (int u = 0; u < 8; u++) r ^= a[u] ^ (uint32_t)acc[u] ^ (uint32_t)(acc[u] >> 32);
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = r;
}
void launch_diag_issue(int kind, unsigned blocks, uint32_t iters, uint32_t* out, unsigned long long* clk, hipStream_t s) {
    if (kind == 0) hipLaunchKernelGGL(k_diag_issue<0>, dim3(blocks), dim3(64), 0, s, iters, out, clk);
    else if (kind == 1) hipLaunchKernelGGL(k_diag_issue<1>, dim3(blocks), dim3(64), 0, s, iters, out, clk);
    else hipLaunchKernelGGL(k_diag_issue<2>, dim3(blocks), dim3(64), 0, s, iters, out, clk);
}

void launch_diag_mulmod(int kind, unsigned blocks, uint32_t iters, uint32_t* out, unsigned long long* clk, hipStream_t s) {
    if (kind == 0) hipLaunchKernelGGL(k_diag_mulmod<0>, dim3(blocks), dim3(64), 0, s, iters, out, clk);
    else if (kind == 1) hipLaunchKernelGGL(k_diag_mulmod<1>, dim3(blocks), dim3(64), 0, s, iters, out, clk);
    else if (kind == 2) hipLaunchKernelGGL(k_diag_mulmod<2>, dim3(blocks), dim3(64), 0, s, iters, out, clk);
    else if (kind == 3) hipLaunchKernelGGL(k_diag_mulmod<3>, dim3(blocks), dim3(64), 0, s, iters, out, clk);
    else hipLaunchKernelGGL(k_diag_mulmod<4>, dim3(blocks), dim3(64), 0, s, iters, out, clk);
}

}  // namespace zkv
