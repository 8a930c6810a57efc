// Stage G2CHK: order-r subgroup membership of B (EIP-197 validation inside the ecPairing precompile).
#include "zkv_internal.h"

namespace zkv {

__global__ __launch_bounds__(ZKV_BLOCK) void k_g2chk(size_t n, Workspace ws, uint8_t* __restrict__ status) {
    size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t flags = ws.flags[i];
    if (!(flags & FL_ALIVE) || (flags & FL_B_INF)) return;
    Fp2 bx, by;
    bx.c0 = ws_ld(ws.prep, ws.cap, 32, i); bx.c1 = ws_ld(ws.prep, ws.cap, 40, i);
    by.c0 = ws_ld(ws.prep, ws.cap, 48, i); by.c1 = ws_ld(ws.prep, ws.cap, 56, i);
    if (!g2_in_subgroup(bx, by)) { ws.g2bad[i] = 1; status[i] = ST_VERIFICATION_FAILED; }
}

void launch_g2chk(size_t n, const Workspace& ws, uint8_t* status, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_g2chk, dim3((unsigned)((n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n, ws, status);
}

}  // namespace zkv
