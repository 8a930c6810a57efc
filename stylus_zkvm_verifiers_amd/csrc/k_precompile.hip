// Precompile-level batch entry points -- the inner seam of the reference: the three EVM precompiles it calls
// (/root/reference/contracts/src/common/groth16.rs:12-14): 0x06 ecAdd (:55), 0x07 ecMul (:54), 0x08 ecPairing (:121-125).
// EIP-196/197 semantics: every coordinate < Q, G1 on curve or (0,0), G2 on the twist AND in the order-r subgroup or
// all-zero; any violation fails the call (ok = 0); pairs containing infinity contribute 1.  One call per lane.
#include "zkv_internal.h"
#include "zkv_scalar.h"     // glv_split / g1_mul_glv

namespace zkv {

__device__ __forceinline__ bool rd_g1(const uint8_t* p, Fp& x, Fp& y, bool& inf) {
    uint32_t a[8], b[8];
    load_be256(a, p); load_be256(b, p + 32);
    if (!raw_lt_p(a) || !raw_lt_p(b)) return false;
    if (raw_is_zero(a) && raw_is_zero(b)) { inf = true; x = fp_zero(); y = fp_zero(); return true; }
    x = fp_from_raw(a); y = fp_from_raw(b); inf = false;
    return g1_on_curve(x, y);
}
__device__ __forceinline__ void wr_be256(uint8_t* p, const Fp& a) {
    uint32_t r[8];
    fp_to_raw(r, a);
#pragma unroll 1
    for (int i = 0; i < 8; i++) {
        uint32_t v = r[7 - i];
        p[4 * i] = (uint8_t)(v >> 24); p[4 * i + 1] = (uint8_t)(v >> 16); p[4 * i + 2] = (uint8_t)(v >> 8); p[4 * i + 3] = (uint8_t)v;
    }
}
__device__ __forceinline__ void wr_g1(uint8_t* out, const G1J& p) {
    G1A a; uint32_t inf;
    g1j_to_affine(p, a, inf);
    wr_be256(out, a.x); wr_be256(out + 32, a.y);          // infinity -> (0,0)
}

__global__ __launch_bounds__(ZKV_BLOCK) void k_ecadd(size_t n, const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint8_t* __restrict__ ok) {
    size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = in + 128 * i;
    Fp x1, y1, x2, y2; bool i1, i2;
    bool good = rd_g1(p, x1, y1, i1) && rd_g1(p + 64, x2, y2, i2);
    G1J acc = g1j_infinity();
    if (good) {
        if (!i1) { acc.x = x1; acc.y = y1; acc.z = fp_one(); }
        if (!i2) acc = g1j_add_affine(acc, x2, y2);
    }
    wr_g1(out + 64 * i, acc);
    ok[i] = good ? 1 : 0;
}

__global__ __launch_bounds__(ZKV_BLOCK) void k_ecmul(size_t n, const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint8_t* __restrict__ ok) {
    size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = in + 96 * i;
    Fp x, y; bool inf;
    bool good = rd_g1(p, x, y, inf);
    uint32_t k[8];
    load_be256(k, p + 64);                                   // any 256-bit scalar
    G1J acc = g1j_infinity();
    if (good && !inf) acc = g1_mul_glv(x, y, k);            // (round 3: 256 doublings and additions before)
    wr_g1(out + 64 * i, acc);
    ok[i] = good ? 1 : 0;
}

// ecPairing: k_pairing_check + k_pairing_miller + k_pairing_finalexp in k_pair.hip (lane-pair kernels, shared with the verify path).

void launch_ecadd(size_t n, const uint8_t* in, uint8_t* out, uint8_t* ok, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_ecadd, dim3((unsigned)((n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n, in, out, ok);
}
void launch_ecmul(size_t n, const uint8_t* in, uint8_t* out, uint8_t* ok, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_ecmul, dim3((unsigned)((n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n, in, out, ok);
}

}  // namespace zkv
