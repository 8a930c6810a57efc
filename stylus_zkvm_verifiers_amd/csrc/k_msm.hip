// Stage MSM: vk_x = base + sum s_b * IC_b via 4-bit fixed-base windows (tables stay L2-resident), then the
// x/y, 1/y normalisation of A', vk_x and C with a single field inversion per proof.
#include "zkv_internal.h"

namespace zkv {

__global__ __launch_bounds__(ZKV_BLOCK) void k_msm(size_t n, const VkTables* __restrict__ vk, Workspace ws) {
    size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t flags = ws.flags[i];
    if (!(flags & FL_ALIVE)) return;
    PrepOut in;
    in.ax = ws_ld(ws.prep, ws.cap, 0, i); in.ay = ws_ld(ws.prep, ws.cap, 8, i);
    in.cx = ws_ld(ws.prep, ws.cap, 16, i); in.cy = ws_ld(ws.prep, ws.cap, 24, i);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        in.s[0][k] = ws.prep[(size_t)(64 + k) * ws.cap + i];
        in.s[1][k] = ws.prep[(size_t)(72 + k) * ws.cap + i];
    }
    G1Norm o;
    msm_normalize(*vk, in, flags, o);
    ws_st(ws.norm, ws.cap, 0, i, o.axs); ws_st(ws.norm, ws.cap, 8, i, o.ays);
    ws_st(ws.norm, ws.cap, 16, i, o.lxs); ws_st(ws.norm, ws.cap, 24, i, o.lys);
    ws_st(ws.norm, ws.cap, 32, i, o.cxs); ws_st(ws.norm, ws.cap, 40, i, o.cys);
    ws.flags[i] = flags;
}

void launch_msm(size_t n, const VkTables* d_tab, const Workspace& ws, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_msm, dim3((unsigned)((n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n, d_tab, ws);
}

}  // namespace zkv
