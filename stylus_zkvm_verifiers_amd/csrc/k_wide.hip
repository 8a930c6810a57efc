// Small-batch variants of the Miller loop and the final exponentiation: ONE PROOF PER 16 LANES (four proofs per
// wavefront), the six Fp2 coefficients of every Fp12 value spread over six lane pairs (csrc/zkv_tower_wide.h).  Selected
// by the C ABI for chunks of at most ZKV_WIDE_BELOW proofs (default 8,192), where the lane-pair kernels cannot fill the
// chip anyway: the answer is the same, a proof just finishes sooner.
#define ZKV_PAIRED 1
#include "zkv_internal.h"
#include "zkv_tower_wide.h"

namespace zkv {

// SLICES = 1: 16 lanes per proof, four proofs per wavefront (chunks of at most ZKV_WIDE_BELOW proofs).
// SLICES = 4: 64 lanes per proof -- one proof per wavefront, the mapping BASELINE.json's north star names -- for chunks of at most
// ZKV_WAVE_BELOW proofs (default 1,024: one wavefront per SIMD), where nothing but the proof's own latency is left to win.
struct WideLane { size_t i; uint32_t g, half; WL w; };
template <int SLICES> __device__ __forceinline__ WideLane wide_lane() {
    constexpr int GROUP = 16 * SLICES, PER_BLOCK = ZKV_BLOCK / GROUP;
    WideLane w;
    w.g = threadIdx.x / GROUP;
    w.half = threadIdx.x & 1u;
    w.w.q = (int)((threadIdx.x >> 1) & 7u);
    if (w.w.q >= 6) w.w.q -= 6;                         // pairs 6 and 7 shadow pairs 0 and 1
    w.w.s = (int)((threadIdx.x % GROUP) >> 4);
    w.i = (size_t)blockIdx.x * PER_BLOCK + w.g;
    return w;
}
__device__ __forceinline__ Fp2 ld_b_w(const Workspace& ws, int word0, size_t i, uint32_t half) {
    Fp2 r; r.h = ws_ld(ws.prep, ws.cap, word0 + 8 * (int)half, i);
    return r;
}

template <int SLICES> __device__ __forceinline__ void miller_w_body(size_t n, const VkTables* __restrict__ vk, const Workspace& ws, uint32_t* lds) {
    constexpr int SLOT = 96 + 48 + 13 * 16 + (SLICES > 1 ? W_RED_WORDS : 0);   // per proof: f (6 Fp2), T (3 Fp2), 13 scratch Fp2, the slices' rows (+ their ninth limbs)
    const WideLane w = wide_lane<SLICES>();
    if (w.i >= n) return;
    const uint32_t flags = ws.flags[w.i];
    if (!(flags & FL_ALIVE)) return;        // the subgroup check of B may still be running: its verdict is read by the final exponentiation
    G1Norm nm;
    nm.axs = ws_ld(ws.norm, ws.cap, 0, w.i); nm.ays = ws_ld(ws.norm, ws.cap, 8, w.i);
    nm.lxs = ws_ld(ws.norm, ws.cap, 16, w.i); nm.lys = ws_ld(ws.norm, ws.cap, 24, w.i);
    nm.cxs = ws_ld(ws.norm, ws.cap, 32, w.i); nm.cys = ws_ld(ws.norm, ws.cap, 40, w.i);
    Fp2 bx = ld_b_w(ws, 32, w.i, w.half), by = ld_b_w(ws, 48, w.i, w.half);
    uint32_t* base = lds + w.g * SLOT + 8 * w.half;
    MRef fm = m_ref(base, 1, 16), tm = m_ref(base + 96, 1, 16), sc = m_ref(base + 144, 1, 16), red = m_ref(base + 352, 1, 16);
    miller_loop_w<SLICES>(*vk, flags, nm, bx, by, fm, tm, sc, w.w, red);
    MRef ab = m_ref((uint32_t*)(vk->f_alpha_beta) + 8 * w.half, 1, 16);
    MRef out = m_ref(ws.f + (size_t)(8 * w.half) * ws.cap + w.i, (uint32_t)ws.cap, 16);
    w12_mul<SLICES>(out, fm, ab, w.w, false, red);
}
template <int SLICES> __device__ __forceinline__ void finalexp_w_body(size_t n, const Workspace& ws, uint8_t* __restrict__ status, uint32_t* lds) {
    constexpr int SLOT = 96 + (SLICES > 1 ? W_RED_WORDS : 0);
    const WideLane w = wide_lane<SLICES>();
    if (w.i >= n) return;
    const uint32_t flags = ws.flags[w.i];
    if (!(flags & FL_ALIVE) || ws.g2bad[w.i]) return;
    const uint32_t st = (uint32_t)ws.cap;
    MRef acc = m_ref(lds + w.g * SLOT + 8 * w.half, 1, 16), red = m_ref(lds + w.g * SLOT + 96 + 8 * w.half, 1, 16);
    MRef F = m_ref(ws.f + (size_t)(8 * w.half) * ws.cap + w.i, st, 16);
    MRef E = m_ref(ws.fe + (size_t)(8 * w.half) * ws.cap + w.i, st, 16);
    const bool one = final_exp_is_one_w<SLICES>(F, E, m_off(E, 96), m_off(E, 192), m_off(E, 288), m_off(E, 384), acc, w.w, red);
    if ((threadIdx.x & (16 * SLICES - 1)) == 0) status[w.i] = one ? ST_OK : ST_VERIFICATION_FAILED;
}

__global__ __launch_bounds__(ZKV_BLOCK, 2) void k_miller_w(size_t n, const VkTables* __restrict__ vk, Workspace ws) {
    __shared__ uint32_t lds[4 * (96 + 48 + 13 * 16)];
    miller_w_body<1>(n, vk, ws, lds);
}
__global__ __launch_bounds__(ZKV_BLOCK, 2) void k_finalexp_w(size_t n, Workspace ws, uint8_t* __restrict__ status) {
    __shared__ uint32_t lds[4 * 96];
    finalexp_w_body<1>(n, ws, status, lds);
}
// one proof per wavefront.  (Launch bounds as for the 16-lane kernels: with a larger register budget here the non-inlined inversion both
// final exponentiations call is compiled into AGPRs and k_finalexp_w drops to one wavefront per SIMD -- 2.4 instead of 1.8 ms at 8,192 proofs.)
__global__ __launch_bounds__(ZKV_BLOCK, 2) void k_miller_w64(size_t n, const VkTables* __restrict__ vk, Workspace ws) {
    __shared__ uint32_t lds[96 + 48 + 13 * 16 + W_RED_WORDS];
    miller_w_body<4>(n, vk, ws, lds);
}
__global__ __launch_bounds__(ZKV_BLOCK, 2) void k_finalexp_w64(size_t n, Workspace ws, uint8_t* __restrict__ status) {
    __shared__ uint32_t lds[96 + W_RED_WORDS];
    finalexp_w_body<4>(n, ws, status, lds);
}

// A consumer wavefront that gave up waiting for its producer (miller_lines_wait's spin bound: about half a second -- only a wavefront that died can
// cause it) fails CLOSED: the proof is reported as failing, never as accepted.  So that such an event is not mistaken for a verdict on the proof,
// it is also counted here; zkv_diag_wait_faults (C ABI) reads the counter, and the tests and the bench assert that it stays zero.
__device__ unsigned int g_zkv_wait_faults = 0;
int read_wait_faults(unsigned long long* out) {
    unsigned int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_zkv_wait_faults), sizeof v) != hipSuccess) { (void)hipGetLastError(); return -1; }
    *out = v;
    return 0;
}

// Two wavefronts per proof (workgroup of 128 lanes): wavefront 1 steps the running point and tabulates the line coefficients, wavefront 0
// accumulates f (miller_lines_producer / miller_loop_consumer, zkv_tower_wide.h).  For chunks of at most ZKV_DUAL_BELOW proofs.
__global__ __launch_bounds__(128, 2) void k_miller_w64d(size_t n, const VkTables* __restrict__ vk, Workspace ws, uint8_t* __restrict__ status) {
    constexpr int F_WORDS = 96 + 64 + W_RED_WORDS, T_WORDS = 48 + 13 * 16, LINE_WORDS = ZKV_MILLER_STEPS * 48;
    __shared__ uint32_t lds[F_WORDS + T_WORDS + LINE_WORDS + 4];
    const size_t i = blockIdx.x;
    if (i >= n) return;
    const uint32_t flags = ws.flags[i];
    if (!(flags & FL_ALIVE)) return;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lane = threadIdx.x & 63u, half = lane & 1u;
    WL w;
    w.q = (int)((lane >> 1) & 7u);
    if (w.q >= 6) w.q -= 6;
    w.s = (int)(lane >> 4);
    volatile uint32_t* ready = lds + F_WORDS + T_WORDS + LINE_WORDS;
    MRef lines = m_ref(lds + F_WORDS + T_WORDS + 8 * half, 1, 16);
    const bool do_ab = !(flags & (FL_A_INF | FL_B_INF));
    // the counter starts at 0: no barrier is needed for that (LDS is not zeroed), so wavefront 1 clears it and wavefront 0 waits for the
    // workgroup barrier below -- the only one in the kernel, reached by both wavefronts before anything else
    if (threadIdx.x == 64) *ready = 0;
    __syncthreads();
    if (wave == 1) {
        if (!do_ab) return;
        const Fp2 bx = ld_b_w(ws, 32, i, half), by = ld_b_w(ws, 48, i, half);
        MRef tm = m_ref(lds + F_WORDS + 8 * half, 1, 16), sc = m_ref(lds + F_WORDS + 48 + 8 * half, 1, 16);
        miller_lines_producer(bx, by, tm, sc, lines, ready, w.q);
        return;
    }
    G1Norm nm;
    nm.axs = ws_ld(ws.norm, ws.cap, 0, i); nm.ays = ws_ld(ws.norm, ws.cap, 8, i);
    nm.lxs = ws_ld(ws.norm, ws.cap, 16, i); nm.lys = ws_ld(ws.norm, ws.cap, 24, i);
    nm.cxs = ws_ld(ws.norm, ws.cap, 32, i); nm.cys = ws_ld(ws.norm, ws.cap, 40, i);
    uint32_t* base = lds + 8 * half;
    MRef fm = m_ref(base, 1, 16), sc = m_ref(base + 96, 1, 16), red = m_ref(base + 160, 1, 16);
    if (!miller_loop_consumer<4>(vk, flags, nm, fm, sc, lines, ready, w, red)) {
        if (lane == 0) { ws.g2bad[i] = 1; status[i] = ST_VERIFICATION_FAILED; atomicAdd(&g_zkv_wait_faults, 1u); }     // unreachable unless the producer wavefront died: fail closed, and count it
        return;
    }
    MRef ab = m_ref((uint32_t*)(vk->f_alpha_beta) + 8 * half, 1, 16);
    MRef out = m_ref(ws.f + (size_t)(8 * half) * ws.cap + i, (uint32_t)ws.cap, 16);
    w12_mul<4>(out, fm, ab, w, false, red);
}
void launch_miller_w64d(size_t n, const VkTables* d_tab, const Workspace& ws, uint8_t* status, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_miller_w64d, dim3((unsigned)n), dim3(128), 0, s, n, d_tab, ws, status);
}

// The ecPairing precompile for SMALL batches of calls (the reference makes exactly one such call per proof: common/groth16.rs:109-128), on
// the two-wavefront Miller kernel: one launch per pair index j as in k_pairing_check + k_pairing_miller (k_pair.hip), but one CALL per workgroup of two
// wavefronts instead of one call per lane pair -- a single 4-pair call takes 4 x 0.8 + 0.8 ms instead of 4 x 5 + 3 ms.  The producer
// wavefront's final running point gives the subgroup verdict for Q (miller_point_closes), as in the lane-pair kernels.
ZKV_HD bool pair_all_w(bool mine) {
    uint32_t v = mine ? 0u : 1u;
    v |= zkv_partner_u32(v);
    return v == 0;
}
__global__ __launch_bounds__(128, 2) void k_pairing_pair_w64d(size_t n, uint32_t k, uint32_t j, const uint8_t* __restrict__ in, Workspace ws, uint8_t* __restrict__ ok) {
    constexpr int F_WORDS = 96 + 64 + W_RED_WORDS, T_WORDS = 48 + 13 * 16, LINE_WORDS = ZKV_MILLER_STEPS * 48;
    __shared__ uint32_t lds[F_WORDS + T_WORDS + LINE_WORDS + 4];
    const size_t i = blockIdx.x;
    if (i >= n) return;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lane = threadIdx.x & 63u, half = lane & 1u;
    WL w;
    w.q = (int)((lane >> 1) & 7u);
    if (w.q >= 6) w.q -= 6;
    w.s = (int)(lane >> 4);
    volatile uint32_t* ready = lds + F_WORDS + T_WORDS + LINE_WORDS;
    volatile uint32_t* verdict = ready + 1;
    MRef lines = m_ref(lds + F_WORDS + T_WORDS + 8 * half, 1, 16);
    MRef P = m_ref(ws.f + (size_t)(8 * half) * ws.cap + i, (uint32_t)ws.cap, 16);
    const bool earlier_ok = j == 0 || ok[i] != 0;         // read before the barrier: wavefront 0 may clear ok[i] after it
    if (threadIdx.x == 64) { *ready = 0; *verdict = 0; }
    __syncthreads();
    if (!earlier_ok) return;                              // an earlier pair of this call was invalid (uniform over the workgroup)
    if (j == 0 && wave == 0) { w12_set_one(P, w.q); if (lane == 0) ok[i] = 1; }
    const uint8_t* p = in + (size_t)192 * ((size_t)k * i + j);
    uint32_t gx[8], gy[8], qxw[8], qyw[8];
    load_be256(gx, p); load_be256(gy, p + 32);
    load_be256(qxw, p + 64 + 32 * (1 - half));             // wire order (imaginary, real): the even lane takes the real parts
    load_be256(qyw, p + 128 + 32 * (1 - half));
    bool okj = raw_lt_p(gx) && raw_lt_p(gy);
    okj = pair_all_w(okj && raw_lt_p(qxw) && raw_lt_p(qyw));
    const bool pinf = raw_is_zero(gx) && raw_is_zero(gy);
    const bool qinf = pair_all_w(raw_is_zero(qxw) && raw_is_zero(qyw));
    Fp px = fp_zero(), py = fp_zero();
    if (okj && !pinf) { px = fp_from_raw(gx); py = fp_from_raw(gy); okj = g1_on_curve(px, py); }
    bool run = false;
    Fp2 qx, qy; qx.h = fp_zero(); qy.h = fp_zero();
    if (okj && !qinf) {
        qx.h = fp_from_raw(qxw); qy.h = fp_from_raw(qyw);
        okj = g2_on_twist(qx, qy);
        run = okj;
    }
    // every lane of both wavefronts has computed the same okj / run from the same bytes
    if (wave == 1) {
        if (!run) return;
        MRef tm = m_ref(lds + F_WORDS + 8 * half, 1, 16), sc = m_ref(lds + F_WORDS + 48 + 8 * half, 1, 16);
        miller_lines_producer(qx, qy, tm, sc, lines, ready, w.q);
        const bool in_g2 = miller_point_closes(tm, qx, qy);        // the loop is the subgroup test of Q as well (miller_loop_p)
        if (lane == 0) *verdict = in_g2 ? 1u : 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) *ready = (uint32_t)(ZKV_MILLER_STEPS + 1);
        return;
    }
    if (run) {
        G1Norm nm;
        const Fp iy = pinf ? fp_zero() : fp_inv(py);
        nm.axs = fp_mul(px, iy); nm.ays = iy;
        nm.lxs = nm.lys = nm.cxs = nm.cys = fp_zero();
        uint32_t* base = lds + 8 * half;
        MRef fm = m_ref(base, 1, 16), sc = m_ref(base + 96, 1, 16), red = m_ref(base + 160, 1, 16);
        // for P = infinity only the point is stepped (no line products): the pair contributes 1 but Q is still judged
        bool fine = miller_loop_consumer<4>((const VkTables*)nullptr, pinf ? (uint32_t)FL_A_INF : 0u, nm, fm, sc, lines, ready, w, red);
        fine = miller_lines_wait(ready, (uint32_t)(ZKV_MILLER_STEPS + 1)) && fine;
        if (!fine && lane == 0) atomicAdd(&g_zkv_wait_faults, 1u);          // the producer vanished (see g_zkv_wait_faults): the call fails closed
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        okj = fine && *verdict != 0;
        if (okj && !pinf) w12_mul<4>(P, P, fm, w, false, red);
    }
    if (!okj && lane == 0) ok[i] = 0;
}
__global__ __launch_bounds__(ZKV_BLOCK, 2) void k_pairing_finalexp_w64(size_t n, Workspace ws, const uint8_t* __restrict__ ok, uint8_t* __restrict__ result, uint32_t empty) {
    __shared__ uint32_t lds[96 + W_RED_WORDS];
    const size_t i = blockIdx.x;
    if (i >= n) return;
    const uint32_t lane = threadIdx.x & 63u, half = lane & 1u;
    WL w;
    w.q = (int)((lane >> 1) & 7u);
    if (w.q >= 6) w.q -= 6;
    w.s = (int)(lane >> 4);
    uint8_t res = empty ? 1 : 0;                                 // k = 0: the empty product is 1
    if (ok[i] && !empty) {
        const uint32_t st = (uint32_t)ws.cap;
        MRef acc = m_ref(lds + 8 * half, 1, 16), red = m_ref(lds + 96 + 8 * half, 1, 16);
        MRef F = m_ref(ws.f + (size_t)(8 * half) * ws.cap + i, st, 16);
        MRef E = m_ref(ws.fe + (size_t)(8 * half) * ws.cap + i, st, 16);
        res = final_exp_is_one_w<4>(F, E, m_off(E, 96), m_off(E, 192), m_off(E, 288), m_off(E, 384), acc, w, red) ? 1 : 0;
    }
    if (lane == 0) result[i] = res;
}
void launch_pairing_w(size_t n, uint32_t k, const uint8_t* in, const Workspace& ws, uint8_t* result, uint8_t* ok, hipStream_t s) {
    if (!n) return;
    if (k == 0) (void)hipMemsetAsync(ok, 1, n, s);
    for (uint32_t j = 0; j < k; j++) hipLaunchKernelGGL(k_pairing_pair_w64d, dim3((unsigned)n), dim3(128), 0, s, n, k, j, in, ws, ok);
    hipLaunchKernelGGL(k_pairing_finalexp_w64, dim3((unsigned)n), dim3(ZKV_BLOCK), 0, s, n, ws, ok, result, k == 0 ? 1u : 0u);
}

static inline unsigned wide_grid(size_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }
void launch_miller_w(size_t n, const VkTables* d_tab, const Workspace& ws, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_miller_w, dim3(wide_grid(n, 4)), dim3(ZKV_BLOCK), 0, s, n, d_tab, ws);
}
void launch_finalexp_w(size_t n, const Workspace& ws, uint8_t* status, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_finalexp_w, dim3(wide_grid(n, 4)), dim3(ZKV_BLOCK), 0, s, n, ws, status);
}
void launch_miller_w64(size_t n, const VkTables* d_tab, const Workspace& ws, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_miller_w64, dim3(wide_grid(n, 1)), dim3(ZKV_BLOCK), 0, s, n, d_tab, ws);
}
void launch_finalexp_w64(size_t n, const Workspace& ws, uint8_t* status, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_finalexp_w64, dim3(wide_grid(n, 1)), dim3(ZKV_BLOCK), 0, s, n, ws, status);
}

}  // namespace zkv
