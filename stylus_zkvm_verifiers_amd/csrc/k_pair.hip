// Lane-pair variants of the three Fp2-heavy stages (G2 subgroup check, Miller loop, final exponentiation):
// ONE PROOF PER PAIR OF ADJACENT LANES, 32 proofs per wavefront.  The even lane holds the real component of every
// Fp2 value and the odd lane the imaginary one; Fp2 products and squarings exchange operands with a DPP quad
// permute (v_mov_b32_dpp quad_perm:[1,0,3,2]) and each lane performs two 81-term column products and one Montgomery
// reduction.  Halving the per-lane tower state (Fp12 = 48 words per lane) brings the kernels under 256 VGPRs and the
// LDS slots under 20 KiB per wave, so two wavefronts are resident per SIMD -- and a 2^16-proof batch already
// supplies them (2^17 lanes = 2048 waves on 1024 SIMDs).
#define ZKV_PAIRED 1
#include "zkv_internal.h"

namespace zkv {

__device__ __forceinline__ Fp2 ld_b(const Workspace& ws, int word0, size_t i) {
    Fp2 r; r.h = ws_ld(ws.prep, ws.cap, word0 + 8 * (int)(threadIdx.x & 1u), i);
    return r;
}

__global__ __launch_bounds__(ZKV_BLOCK, 2) void k_g2chk2(size_t n, Workspace ws, uint8_t* __restrict__ status) {
    size_t i = ((size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x) >> 1;
    if (i >= n) return;
    uint32_t flags = ws.flags[i];
    if (!(flags & FL_ALIVE) || (flags & FL_B_INF)) return;
    Fp2 bx = ld_b(ws, 32, i), by = ld_b(ws, 48, i);
    bool ok = g2_in_subgroup(bx, by);
    if (!ok && !(threadIdx.x & 1u)) { ws.g2bad[i] = 1; status[i] = ST_VERIFICATION_FAILED; }
}

__global__ __launch_bounds__(ZKV_BLOCK, 2) void k_miller2(size_t n, const VkTables* __restrict__ vk, Workspace ws) {
    __shared__ uint32_t lds[(48 + 24) * ZKV_BLOCK];       // f: 6 Fp per lane, T: 3 Fp per lane, lane-interleaved
    size_t i = ((size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x) >> 1;
    if (i >= n) return;
    uint32_t flags = ws.flags[i];
    if (!(flags & FL_ALIVE)) return;        // the subgroup check of B may still be running: its verdict is read by k_finalexp2
    const uint32_t par = threadIdx.x & 1u;
    LRef fm = l_ref(lds + threadIdx.x);
    LRef tm = l_ref(lds + 48 * ZKV_BLOCK + threadIdx.x);
    SoaRef norm = {ws.norm + i, ws.cap};                                        // Fp values: both lanes of the pair read them
    SoaRef bsrc = {ws.prep + (size_t)(32 + 8 * par) * ws.cap + i, ws.cap};      // this lane's component of B.x (B.y 16 words on)
    miller_loop_p(vk, flags, norm, bsrc, fm, tm);
    MRef ab = m_ref((uint32_t*)(vk->f_alpha_beta) + 8 * par, 1, 16);
    MRef out = m_ref(ws.f + (size_t)(8 * par) * ws.cap + i, (uint32_t)ws.cap, 16);
    f12m_mul_body(out, fm, ab, false);
}

__global__ __launch_bounds__(ZKV_BLOCK, 2) void k_finalexp2(size_t n, Workspace ws, uint8_t* __restrict__ status) {
    __shared__ uint32_t lds[48 * ZKV_BLOCK];
    size_t i = ((size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x) >> 1;
    if (i >= n) return;
    uint32_t flags = ws.flags[i];
    if (!(flags & FL_ALIVE) || ws.g2bad[i]) return;
    const uint32_t par = threadIdx.x & 1u;
    const uint32_t st = (uint32_t)ws.cap;
    LRef acc = l_ref(lds + threadIdx.x);
    MRef F = m_ref(ws.f + (size_t)(8 * par) * ws.cap + i, st, 16);
    MRef E = m_ref(ws.fe + (size_t)(8 * par) * ws.cap + i, st, 16);
    MRef accm = m_ref(lds + threadIdx.x, 64, 8);          // the accumulator's LDS words through a flat pointer, for the rare generic operations
    bool one = final_exp_prog_p(F, E, acc, accm);
    if (!par) status[i] = one ? ST_OK : ST_VERIFICATION_FAILED;
}

static inline unsigned pair_grid(size_t n) { return (unsigned)((2 * n + ZKV_BLOCK - 1) / ZKV_BLOCK); }

void launch_g2chk2(size_t n, const Workspace& ws, uint8_t* status, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_g2chk2, dim3(pair_grid(n)), dim3(ZKV_BLOCK), 0, s, n, ws, status);
}
void launch_miller2(size_t n, const VkTables* d_tab, const Workspace& ws, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_miller2, dim3(pair_grid(n)), dim3(ZKV_BLOCK), 0, s, n, d_tab, ws);
}
void launch_finalexp2(size_t n, const Workspace& ws, uint8_t* status, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_finalexp2, dim3(pair_grid(n)), dim3(ZKV_BLOCK), 0, s, n, ws, status);
}

}  // namespace zkv
