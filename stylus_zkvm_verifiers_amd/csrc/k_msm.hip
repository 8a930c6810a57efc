// Stage MSM: vk_x = base + sum s_b * IC_b via fixed-base windows -- 16-bit ones for the big batches (Msm16: one 64-byte gather and one
// addition per 16 bits of a scalar, the table in HBM / the memory-side cache), the L2-resident 8-bit rows otherwise -- then the
// x/y, 1/y normalisation of A', vk_x and C with a single field inversion per proof.
#include "zkv_internal.h"

namespace zkv {

__global__ __launch_bounds__(ZKV_BLOCK) void k_msm(size_t n, const VkTables* __restrict__ vk, Msm16 m16, const InstTab* __restrict__ inst_tab, Workspace ws) {
    size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t flags = ws.flags[i];
    if (!(flags & FL_ALIVE)) return;
    // the scalars' window digits are read from the proof's workspace row as the walk needs them (L1 / L2 hits: 64 bytes per proof)
    auto word = [&](uint32_t b, uint32_t k) { return ws.prep[(size_t)(64 + 8 * b + k) * ws.cap + i]; };
    const G1A* base = &vk->base; uint32_t base_inf = vk->base_inf;
    if (inst_tab) { const InstTab& t = inst_tab[flags >> 8]; flags &= 0xFFu; base = &t.base; base_inf = t.base_inf; }     // verifier set: the instance index rides in the upper bits of the flags word
    const G1J acc = m16.tab ? msm_accumulate_w16(*vk, m16, word, *base, base_inf) : msm_accumulate_w(*vk, word, *base, base_inf);
    PrepOut in;
    in.ax = ws_ld(ws.prep, ws.cap, 0, i); in.ay = ws_ld(ws.prep, ws.cap, 8, i);
    in.cx = ws_ld(ws.prep, ws.cap, 16, i); in.cy = ws_ld(ws.prep, ws.cap, 24, i);
    G1Norm o;
    msm_normalize_acc(acc, in, flags, o);
    ws_st(ws.norm, ws.cap, 0, i, o.axs); ws_st(ws.norm, ws.cap, 8, i, o.ays);
    ws_st(ws.norm, ws.cap, 16, i, o.lxs); ws_st(ws.norm, ws.cap, 24, i, o.lys);
    ws_st(ws.norm, ws.cap, 32, i, o.cxs); ws_st(ws.norm, ws.cap, 40, i, o.cys);
    ws.flags[i] = flags;
}

// The same stage for very small chunks: ONE PROOF PER WAVEFRONT.  A lane adds the table entries of its windows (SP1: 64 windows, one
// per lane; RISC Zero: 32; a generic key: up to 160, three per lane), a six-round butterfly of complete Jacobian additions sums the 64
// partial points (ds_bpermute exchanges), lane 0 adds the base point and normalises.  About 40 k instructions deep instead of the 200 k
// of a lane walking all windows alone: a single SP1 proof's MSM 0.61 -> see DESIGN ms.  Costs 64 lanes per proof, so only below
// ZKV_MSM_WAVE_BELOW proofs (one round of wavefronts).
__device__ __forceinline__ G1J g1j_shfl_xor(const G1J& p, int mask) {
    G1J r;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        r.x.v[k] = (uint32_t)__shfl_xor((int)p.x.v[k], mask, 64);
        r.y.v[k] = (uint32_t)__shfl_xor((int)p.y.v[k], mask, 64);
        r.z.v[k] = (uint32_t)__shfl_xor((int)p.z.v[k], mask, 64);
    }
    return r;
}
__global__ __launch_bounds__(64) void k_msm_w(size_t n, const VkTables* __restrict__ vk, const InstTab* __restrict__ inst_tab, Workspace ws) {
    const size_t i = blockIdx.x;
    if (i >= n) return;
    uint32_t flags = ws.flags[i];
    if (!(flags & FL_ALIVE)) return;                            // the whole wavefront leaves together
    const uint32_t lane = threadIdx.x, nv = vk->n_var;
    G1J acc = g1j_infinity();
    uint32_t first = 0;
#pragma unroll 1
    for (uint32_t b = 0; b < nv; b++) {
        const uint32_t nw = vk->var_windows[b];
#pragma unroll 1
        for (uint32_t w = (lane + 64u - (first & 63u)) & 63u; w < nw; w += 64) {     // global window index first + w = lane (mod 64)
            const uint32_t d = (ws.prep[(size_t)(64 + 8 * b + (w >> 2)) * ws.cap + i] >> ((w & 3) * 8)) & 255u;
            if (d) { const G1A& e = vk->msm[b][w][d]; acc = g1j_add_affine(acc, e.x, e.y); }
        }
        first += nw;
    }
#pragma unroll 1
    for (int m = 32; m >= 1; m >>= 1) acc = g1j_add(acc, g1j_shfl_xor(acc, m));
    if (lane != 0) return;
    const G1A* base = &vk->base; uint32_t base_inf = vk->base_inf;
    if (inst_tab) { const InstTab& t = inst_tab[flags >> 8]; flags &= 0xFFu; base = &t.base; base_inf = t.base_inf; }
    if (!base_inf) acc = g1j_add_affine(acc, base->x, base->y);
    PrepOut in;
    in.ax = ws_ld(ws.prep, ws.cap, 0, i); in.ay = ws_ld(ws.prep, ws.cap, 8, i);
    in.cx = ws_ld(ws.prep, ws.cap, 16, i); in.cy = ws_ld(ws.prep, ws.cap, 24, i);
    G1Norm o;
    msm_normalize_acc(acc, in, flags, o);
    ws_st(ws.norm, ws.cap, 0, i, o.axs); ws_st(ws.norm, ws.cap, 8, i, o.ays);
    ws_st(ws.norm, ws.cap, 16, i, o.lxs); ws_st(ws.norm, ws.cap, 24, i, o.lys);
    ws_st(ws.norm, ws.cap, 32, i, o.cxs); ws_st(ws.norm, ws.cap, 40, i, o.cys);
    ws.flags[i] = flags;
}
void launch_msm_w(size_t n, const VkTables* d_tab, const InstTab* inst_tab, const Workspace& ws, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_msm_w, dim3((unsigned)n), dim3(64), 0, s, n, d_tab, inst_tab, ws);
}

// compute_vk_x alone (zkv_ctx_vk_x_batch): same tables and window walk as k_msm, affine result as 64 big-endian bytes.
__global__ __launch_bounds__(ZKV_BLOCK) void k_vk_x(size_t n, const VkTables* __restrict__ vk, Msm16 m16, const InstTab* __restrict__ inst_tab,
                                                     const uint32_t* __restrict__ inst, const uint8_t* __restrict__ sig, uint8_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t nv = vk->n_var;                                  // 2 for the RISC Zero / SP1 keys, n_ic - 1 for a generic key
    // limb k (least significant first) of the big-endian 32-byte signal b, read where the window walk needs it
    auto word = [&](uint32_t b, uint32_t k) {
        const uint8_t* q = sig + 32 * ((size_t)nv * i + b) + 4 * (7 - k);
        return ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
    };
    const G1A* base = &vk->base; uint32_t base_inf = vk->base_inf;
    if (inst_tab) { const InstTab& t = inst_tab[inst[i]]; base = &t.base; base_inf = t.base_inf; }
    G1J acc = m16.tab ? msm_accumulate_w16(*vk, m16, word, *base, base_inf) : msm_accumulate_w(*vk, word, *base, base_inf);
    G1A a; uint32_t inf;
    g1j_to_affine(acc, a, inf);
    uint32_t r[8];
    uint8_t* o = out + 64 * i;
#pragma unroll 1
    for (int c = 0; c < 2; c++) {
        fp_to_raw(r, c ? a.y : a.x);
#pragma unroll 1
        for (int k = 0; k < 8; k++) {
            uint32_t v = r[7 - k];
            o[32 * c + 4 * k] = (uint8_t)(v >> 24); o[32 * c + 4 * k + 1] = (uint8_t)(v >> 16);
            o[32 * c + 4 * k + 2] = (uint8_t)(v >> 8); o[32 * c + 4 * k + 3] = (uint8_t)v;
        }
    }
}
void launch_vk_x(size_t n, const VkTables* d_tab, const Msm16& m16, const InstTab* inst_tab, const uint32_t* inst, const uint8_t* sig, uint8_t* out, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_vk_x, dim3((unsigned)((n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n, d_tab, m16, inst_tab, inst, sig, out);
}

void launch_msm(size_t n, const VkTables* d_tab, const Msm16& m16, const InstTab* inst_tab, const Workspace& ws, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_msm, dim3((unsigned)((n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n, d_tab, m16, inst_tab, ws);
}

}  // namespace zkv
