// Stage FINALEXP: f^(k (p^12-1)/r) == 1  <=>  accept.  The hot accumulator lives in LDS (lane-interleaved),
// the five cold Fp12 values in HBM struct-of-arrays slots (coalesced, touched once per ~60 squarings).
#include "zkv_internal.h"

namespace zkv {

__global__ __launch_bounds__(ZKV_BLOCK) void k_finalexp(size_t n, Workspace ws, uint8_t* __restrict__ status) {
    __shared__ uint32_t lds[96 * ZKV_BLOCK];
    size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t flags = ws.flags[i];
    if (!(flags & FL_ALIVE) || ws.g2bad[i]) return;
    uint32_t st = (uint32_t)ws.cap;
    LRef acc = l_ref(lds + threadIdx.x);
    MRef F = m_ref(ws.f + i, st);
    MRef E = m_ref(ws.fe + i, st);
    MRef Y1 = m_off(E, 96), Y3 = m_off(E, 192), Y4 = m_off(E, 288);
    status[i] = final_exp_is_one_m(F, E, Y1, Y3, Y4, m_off(E, 384), acc) ? ST_OK : ST_VERIFICATION_FAILED;
}

void launch_finalexp(size_t n, const Workspace& ws, uint8_t* status, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_finalexp, dim3((unsigned)((n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n, ws, status);
}

}  // namespace zkv
