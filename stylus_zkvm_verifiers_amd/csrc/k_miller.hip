// Stage MILLER: shared-accumulator optimal-ate Miller loop of e(A',B) e(vk_x,gamma) e(C,delta), times the
// precomputed Miller value of (alpha,beta).  One proof per lane; the Fp12 accumulator f and the running G2
// point T of every lane live in LDS, lane-interleaved (word k of lane l at lds[k*64 + l]: conflict-free
// ds_read/ds_write_b32), gamma/delta line coefficients are wave-uniform table reads.
#include "zkv_internal.h"

namespace zkv {

__global__ __launch_bounds__(ZKV_BLOCK) void k_miller(size_t n, const VkTables* __restrict__ vk, Workspace ws) {
    __shared__ uint32_t lds[(96 + 48) * ZKV_BLOCK];
    size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t flags = ws.flags[i];
    if (!(flags & FL_ALIVE)) return;        // the subgroup check of B may still be running: its verdict is read by the final exponentiation
    G1Norm nm;
    nm.axs = ws_ld(ws.norm, ws.cap, 0, i); nm.ays = ws_ld(ws.norm, ws.cap, 8, i);
    nm.lxs = ws_ld(ws.norm, ws.cap, 16, i); nm.lys = ws_ld(ws.norm, ws.cap, 24, i);
    nm.cxs = ws_ld(ws.norm, ws.cap, 32, i); nm.cys = ws_ld(ws.norm, ws.cap, 40, i);
    Fp2 bx, by;
    bx.c0 = ws_ld(ws.prep, ws.cap, 32, i); bx.c1 = ws_ld(ws.prep, ws.cap, 40, i);
    by.c0 = ws_ld(ws.prep, ws.cap, 48, i); by.c1 = ws_ld(ws.prep, ws.cap, 56, i);
    LRef fm = l_ref(lds + threadIdx.x);
    LRef tm = l_ref(lds + 96 * ZKV_BLOCK + threadIdx.x);
    miller_loop_m(vk, flags, nm, bx, by, fm, tm);
    MRef ab = m_ref((uint32_t*)(vk->f_alpha_beta), 1);
    MRef out = m_ref(ws.f + i, (uint32_t)ws.cap);
    f12m_mul(out, fm, ab);
}

void launch_miller(size_t n, const VkTables* d_tab, const Workspace& ws, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_miller, dim3((unsigned)((n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n, d_tab, ws);
}

}  // namespace zkv
