// Context set-up kernels: derive the device tables of one verification key (run once per context).
#include "zkv_internal.h"

namespace zkv {

// fixed part of vk_x: IC[0] + sum of per-context signals
__global__ __launch_bounds__(64) void k_setup_base(const VkRaw* __restrict__ raw, VkTables* __restrict__ t) {
    if (threadIdx.x == 0) setup_base(*raw, *t);
}
// precompile-input validity of every VK point (a generic key may be anything)
__global__ __launch_bounds__(64) void k_setup_validate(const VkRaw* __restrict__ raw, VkTables* __restrict__ t) {
    if (threadIdx.x == 0) setup_validate(*raw, *t);
}
// gamma / delta slope-line tables for the fixed-Q Miller loops
__global__ __launch_bounds__(64) void k_setup_lines(const VkRaw* __restrict__ raw, VkTables* __restrict__ t) {
    if (threadIdx.x == 0) setup_lines(raw->gamma, t->lines[0]);
    if (threadIdx.x == 1) setup_lines(raw->delta, t->lines[1]);
}
// one lane per (scalar, window) row of the fixed-base MSM table; all 32 windows of every per-proof scalar (the vk_x stage reads
// var_windows[b] of them, the aggregate check all)
__global__ __launch_bounds__(64) void k_setup_msm(const VkRaw* __restrict__ raw, VkTables* __restrict__ t) {
    int b = blockIdx.x, w = threadIdx.x;
    if (b < (int)raw->n_var && w < MSM_MAX_WINDOWS) setup_msm_row(*raw, *t, b, w);
}
// Miller value of (alpha, beta); f and T of the single lane live in LDS like in k_miller
__global__ __launch_bounds__(64) void k_setup_alpha_beta(const VkRaw* __restrict__ raw, VkTables* __restrict__ t) {
    __shared__ uint32_t lds[96 + 48];
    if (threadIdx.x != 0) return;
    MRef fm = m_ref(lds, 1);
    MRef tm = m_ref(lds + 96, 1);
    setup_alpha_beta(*raw, *t, fm, tm);
}

// one lane per instance of a verifier set: selector (SHA-256 on the device), control-id range check, per-instance part of vk_x
__global__ __launch_bounds__(64) void k_setup_instances(const VkRaw* __restrict__ raw, InstConsts k, const InstRaw* __restrict__ in, InstTab* __restrict__ out,
                                                        uint32_t n_inst) {
    uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i < n_inst) setup_instance(*raw, k, in[i], out[i]);
}
void launch_setup_instances(const VkRaw* d_raw, const InstConsts& k, const InstRaw* d_in, InstTab* d_out, uint32_t n_inst, hipStream_t s) {
    if (!n_inst) return;
    hipLaunchKernelGGL(k_setup_instances, dim3((n_inst + 63) / 64), dim3(64), 0, s, d_raw, k, d_in, d_out, n_inst);
}

// wide vk_x windows: one lane per 64 entries (row, upper digit, quarter of the lower digits); after k_setup_msm
__global__ __launch_bounds__(64) void k_setup_msm16(const VkTables* __restrict__ t, Msm16 m, G1A* __restrict__ tab, uint32_t rows) {
    const uint32_t id = blockIdx.x * 64 + threadIdx.x;
    const uint32_t r = id >> 10, hi = (id >> 2) & 255u, q = id & 3u;
    if (r >= rows) return;
    uint32_t b = 0;
    while (b + 1 < t->n_var && m.row0[b + 1] <= r) b++;
    setup_msm16_chunk(*t, tab + ((size_t)r << 16), b, r - m.row0[b], hi, 64 * q);
}
void launch_setup_msm16(const VkTables* d_tab, const Msm16& m, G1A* tab, uint32_t rows, hipStream_t s) {
    if (!rows) return;
    hipLaunchKernelGGL(k_setup_msm16, dim3(rows * 16), dim3(64), 0, s, d_tab, m, tab, rows);
}

void launch_setup(const VkRaw* d_raw, VkTables* d_tab, hipStream_t s) {
    hipLaunchKernelGGL(k_setup_validate, dim3(1), dim3(64), 0, s, d_raw, d_tab);
    hipLaunchKernelGGL(k_setup_base, dim3(1), dim3(64), 0, s, d_raw, d_tab);
    hipLaunchKernelGGL(k_setup_lines, dim3(1), dim3(64), 0, s, d_raw, d_tab);
    hipLaunchKernelGGL(k_setup_msm, dim3(MAX_VAR), dim3(64), 0, s, d_raw, d_tab);
    hipLaunchKernelGGL(k_setup_alpha_beta, dim3(1), dim3(64), 0, s, d_raw, d_tab);
}

}  // namespace zkv
