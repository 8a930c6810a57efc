// Lane-pair variants of the three Fp2-heavy stages (G2 subgroup check, Miller loop, final exponentiation):
// ONE PROOF PER PAIR OF ADJACENT LANES, 32 proofs per wavefront.  The even lane holds the real component of every
// Fp2 value and the odd lane the imaginary one; Fp2 products and squarings exchange operands with a DPP quad
// permute (v_mov_b32_dpp quad_perm:[1,0,3,2]) and each lane performs two 81-term column products and one Montgomery
// reduction.  Halving the per-lane tower state (Fp12 = 48 words per lane) brings the kernels under 256 VGPRs and the
// LDS slots under 20 KiB per wave, so two wavefronts are resident per SIMD -- and a 2^16-proof batch already
// supplies them (2^17 lanes = 2048 waves on 1024 SIMDs).
#define ZKV_PAIRED 1
#include "zkv_internal.h"
#include "zkv_agg.h"

namespace zkv {

// Wavefronts per workgroup of the two hot kernels.  The wavefronts of a group share nothing (each has its own LDS rows, no barrier);
// larger groups only mean fewer workgroups for the dispatcher to place.  Round-3 experiment for the 2^16-proof launch: 13.0 / 12.9 / 12.9 ms
// per step with 1 / 2 / 4 (184.0 / 187.4 / 183.9 at 2^20; profiles/round3_n_waves_per_workgroup_ab.txt) -- no effect, the default stays 1.
#ifndef ZKV_PAIR_WAVES
#define ZKV_PAIR_WAVES 1
#endif
constexpr int PAIR_BLOCK = ZKV_BLOCK * ZKV_PAIR_WAVES;

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

#if defined(ZKV_STAMPS)
// Diagnostic build only (tools/ab_build.py ...:-DZKV_STAMPS, tools/stamp_probe.py): lane 0 of every wavefront of k_miller2 / k_finalexp2 leaves
// the constant 100 MHz clock at its first and last instruction, the shader clock, and where it ran (HW_ID, XCC_ID) in a side buffer no
// result depends on: 8 words per wavefront, kernel k at rows [k * n_waves, (k + 1) * n_waves).
__device__ unsigned long long* g_zkv_stamps = nullptr;
extern "C" __attribute__((visibility("default"))) int zkv_diag_set_stamps(void* dev_ptr) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_zkv_stamps), &dev_ptr, sizeof(dev_ptr));
}
__device__ __forceinline__ void zkv_stamp(unsigned kernel, unsigned slot) {
    if ((threadIdx.x & 63u) != 0 || !g_zkv_stamps) return;
    unsigned long long* row = g_zkv_stamps + ((size_t)kernel * gridDim.x * (blockDim.x >> 6) + (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8;
    row[slot] = __builtin_amdgcn_s_memrealtime();
    row[4 + slot] = __builtin_amdgcn_s_memtime();
    if (slot == 0) { row[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4); row[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20); }
}
#define ZKV_STAMP(k, s) zkv_stamp(k, s)
#else
#define ZKV_STAMP(k, s) ((void)0)
#endif

// The Miller loop is also the subgroup test of B (miller_loop_p, check_b): a proof whose B is outside G2 gets the precompile-failure
// status here and is skipped by k_finalexp2.  (k_g2chk2 remains for the 16-lane kernels of small chunks.)
__global__ __launch_bounds__(PAIR_BLOCK, 2) void k_miller2(size_t n, const VkTables* __restrict__ vk, Workspace ws, uint8_t* __restrict__ status) {
    __shared__ uint32_t lds[(48 + 24) * PAIR_BLOCK];      // per wavefront: f: 6 Fp per lane, T: 3 Fp per lane, lane-interleaved
    ZKV_STAMP(0, 0);
    size_t i = ((size_t)blockIdx.x * PAIR_BLOCK + threadIdx.x) >> 1;
    if (i >= n) return;
    uint32_t flags = ws.flags[i];
    if (!(flags & FL_ALIVE)) return;
    const uint32_t par = threadIdx.x & 1u;
    uint32_t* wl = lds + (threadIdx.x >> 6) * ((48 + 24) * ZKV_BLOCK) + (threadIdx.x & 63u);     // this wavefront's rows, this lane's column
    LRef fm = l_ref(wl);
    LRef tm = l_ref(wl + 48 * ZKV_BLOCK);
    SoaRef norm = {ws.norm, ws.cap, (uint32_t)i * 4u};                                               // Fp values: both lanes of the pair read them
    SoaRef bsrc = {ws.prep + 32 * ws.cap, ws.cap, (uint32_t)(8 * par * ws.cap + i) * 4u};             // this lane's component of B.x (B.y 16 words on)
    if (!miller_loop_p(vk, flags, norm, bsrc, fm, tm, true)) {
        if (!par) { ws.g2bad[i] = 1; status[i] = ST_VERIFICATION_FAILED; }
        return;
    }
    MRef ab = m_ref((uint32_t*)(vk->f_alpha_beta) + 8 * par, 1, 16);
    MRef out = m_ref(ws.f + (size_t)(8 * par) * ws.cap + i, (uint32_t)ws.cap, 16);
    f12m_mul_body(out, fm, ab, false);
    ZKV_STAMP(0, 1);
}

__global__ __launch_bounds__(PAIR_BLOCK, 2) void k_finalexp2(size_t n, Workspace ws, uint8_t* __restrict__ status) {
    __shared__ uint32_t lds[54 * PAIR_BLOCK];             // the accumulator in resident 29-bit limbs: 6 coefficients x 9 words per lane
    ZKV_STAMP(1, 0);
    size_t i = ((size_t)blockIdx.x * PAIR_BLOCK + threadIdx.x) >> 1;
    if (i >= n) return;
    uint32_t flags = ws.flags[i];
    if (!(flags & FL_ALIVE) || ws.g2bad[i]) return;
    const uint32_t par = threadIdx.x & 1u;
    uint32_t* wl = lds + (threadIdx.x >> 6) * (54 * ZKV_BLOCK) + (threadIdx.x & 63u);
    L9Ref acc = l9_ref(wl);
    bool one = final_exp_prog_p(ws.f, ws.fe, ws.cap, (uint32_t)(8 * par * ws.cap + i) * 4u, acc);
    if (!par) status[i] = one ? ST_OK : ST_VERIFICATION_FAILED;
    ZKV_STAMP(1, 1);
}

// The ecPairing precompile as a batch (the inner seam of the reference: common/groth16.rs:109-128 builds k x 192 bytes of
// calldata and STATICCALLs 0x08): one CALL per lane pair, k pairs per call, EIP-197 semantics -- every point of every pair is
// validated (coordinates < Q, G1 on the curve or (0,0), G2 on the twist and in the order-r subgroup or all-zero) whether or not
// the pair is skipped; a pair with a point at infinity contributes 1; result = 1 iff the product of the pairings is 1.
// Same building blocks as the verify path: the flat Miller loop with a single variable pair (no tables), the product kept in
// the call's F slot, the interpreted final exponentiation.  The call's workspace rows (ws.norm, ws.prep) carry x/y, 1/y of the
// current G1 point and the current G2 point, which the loop re-reads where it needs them.
ZKV_HD bool pair_all(bool mine) {          // AND over the two lanes of a pair
    uint32_t v = mine ? 0u : 1u;
    v |= zkv_partner_u32(v);
    return v == 0;
}
// Two launches per pair index j: k_pairing_check validates pair j of every call (coordinates < p, P on the curve, Q on the twist),
// writes the normalised rows and a per-call word in ws.flags (0 = identity pair, nothing to do; 1 = run; 2 = run with P = infinity:
// only the point is stepped), k_pairing_miller runs the Miller loop - the subgroup test of Q as well - and multiplies into the call's
// F slot; ok[] collects the verdict on the inputs.  Then the final exponentiation of the valid calls.  (As one kernel per pair the
// check's inversion and the Miller loop shared a frame of 70 spilled VGPRs; with the pair loop inside as well, 217.)
// slot_off: where pair j's rows go -- 0 (the call's own slot, one pair at a time) or j * n (all pairs of a call resident at once, for
// k_pairing_miller_g).
__global__ __launch_bounds__(ZKV_BLOCK, 2) void k_pairing_check(size_t n, uint32_t k, uint32_t j, size_t slot_off, const uint8_t* __restrict__ in, Workspace ws,
                                                                uint8_t* __restrict__ ok) {
    const size_t call = ((size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x) >> 1;
    if (call >= n) return;
    const size_t i = call + slot_off;
    const uint32_t par = threadIdx.x & 1u;
    if (j == 0) {
        f12m_set_one(m_ref(ws.f + (size_t)(8 * par) * ws.cap + call, (uint32_t)ws.cap, 16));
        if (!par) ok[call] = 1;
    } else if (!ok[call]) { if (!par) ws.flags[i] = 0; return; }   // an earlier pair of this call was invalid (both lanes read the same byte)
    const uint8_t* p = in + (size_t)192 * ((size_t)k * call + j);
    uint32_t gx[8], gy[8], qxw[8], qyw[8];
    load_be256(gx, p); load_be256(gy, p + 32);
    load_be256(qxw, p + 64 + 32 * (1 - par));                   // wire order (imaginary, real): the even lane takes the real parts
    load_be256(qyw, p + 128 + 32 * (1 - par));
    bool okj = raw_lt_p(gx) && raw_lt_p(gy);
    okj = pair_all(okj && raw_lt_p(qxw) && raw_lt_p(qyw));
    const bool pinf = raw_is_zero(gx) && raw_is_zero(gy);
    const bool qinf = pair_all(raw_is_zero(qxw) && raw_is_zero(qyw));
    Fp px = fp_zero(), py = fp_zero();
    if (okj && !pinf) { px = fp_from_raw(gx); py = fp_from_raw(gy); okj = g1_on_curve(px, py); }
    uint32_t run = 0;
    if (okj && !qinf) {
        Fp2 qx, qy; qx.h = fp_from_raw(qxw); qy.h = fp_from_raw(qyw);
        okj = g2_on_twist(qx, qy);
        if (okj) {
            const Fp iy = pinf ? fp_zero() : fp_inv(py);
            ws_st(ws.norm, ws.cap, 0, i, fp_mul(px, iy)); ws_st(ws.norm, ws.cap, 8, i, iy);            // both lanes store the same words
            ws_st(ws.prep, ws.cap, 32 + 8 * (int)par, i, qx.h); ws_st(ws.prep, ws.cap, 48 + 8 * (int)par, i, qy.h);
            run = pinf ? 2u : 1u;
        }
    }
    if (!par) { ws.flags[i] = run; if (!okj) ok[call] = 0; }
}
// All pairs of a call in ONE Miller loop with one accumulator (miller_loop_pg: f is squared once per step for the whole call instead of once per
// pair -- for the reference's four pairs, common/groth16.rs:109-128, 24.0 k instead of 32.8 k multiply-adds per step and lane).  Pair p's rows are in
// slot call + p n (k_pairing_check with slot_off = p n), its running point in the final exponentiation's scratch rows of that slot.  k <= G.
template <int G>
__global__ __launch_bounds__(PAIR_BLOCK, 2) void k_pairing_miller_g(size_t n, uint32_t k, Workspace ws, uint8_t* __restrict__ ok) {
    __shared__ uint32_t lds[48 * PAIR_BLOCK];
    const size_t i = ((size_t)blockIdx.x * PAIR_BLOCK + threadIdx.x) >> 1;
    if (i >= n) return;
    if (!ok[i]) return;                                         // both lanes of the pair read the same byte
    uint32_t mask = 0, abmask = 0;
#pragma unroll
    for (uint32_t p = 0; p < (uint32_t)G; p++) {
        const uint32_t run = p < k ? ws.flags[i + p * n] : 0u;
        if (run) { mask |= 1u << p; if (run == 1u) abmask |= 1u << p; }
    }
    const uint32_t par = threadIdx.x & 1u;
    MRef P = m_ref(ws.f + (size_t)(8 * par) * ws.cap + i, (uint32_t)ws.cap, 16);
    if (!mask) return;                                          // every pair is the identity: the F slot holds 1 (k_pairing_check, j = 0)
    uint32_t* wl = lds + (threadIdx.x >> 6) * (48 * ZKV_BLOCK) + (threadIdx.x & 63u);
    LRef fm = l_ref(wl);
    SoaRef norm = {ws.norm, ws.cap, (uint32_t)i * 4u};
    SoaRef bsrc = {ws.prep + 32 * ws.cap, ws.cap, (uint32_t)(8 * par * ws.cap + i) * 4u};
    SoaRW tq = {ws.fe, ws.cap, (uint32_t)(8 * par * ws.cap + i) * 4u};
    const uint32_t fine = miller_loop_pg<G>(mask, abmask, norm, bsrc, tq, (uint32_t)n * 4u, fm);
    if ((fine & mask) != mask) { if (!par) ok[i] = 0; return; }       // a G2 point outside the subgroup
    f12m_copy(P, fm);
}
__global__ __launch_bounds__(ZKV_BLOCK, 2) void k_pairing_miller(size_t n, Workspace ws, uint8_t* __restrict__ ok) {
    __shared__ uint32_t lds[(48 + 24) * ZKV_BLOCK];
    size_t i = ((size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x) >> 1;
    if (i >= n) return;
    const uint32_t run = ws.flags[i];                           // both lanes of the pair read the same word
    if (!run) return;
    const uint32_t par = threadIdx.x & 1u;
    LRef fm = l_ref(lds + threadIdx.x);
    LRef tm = l_ref(lds + 48 * ZKV_BLOCK + threadIdx.x);
    MRef P = m_ref(ws.f + (size_t)(8 * par) * ws.cap + i, (uint32_t)ws.cap, 16);
    SoaRef norm = {ws.norm, ws.cap, (uint32_t)i * 4u};
    SoaRef bsrc = {ws.prep + 32 * ws.cap, ws.cap, (uint32_t)(8 * par * ws.cap + i) * 4u};
    const bool fine = miller_loop_p((const VkTables*)nullptr, run == 2 ? (uint32_t)FL_A_INF : 0u, norm, bsrc, fm, tm, true);
    if (fine && run == 1) f12m_mul(P, fm, P);                 // not inlined: inside this kernel the product kept 70 VGPRs in scratch
    if (!fine && !par) ok[i] = 0;
}
__global__ __launch_bounds__(ZKV_BLOCK, 2) void k_pairing_finalexp(size_t n, Workspace ws, const uint8_t* __restrict__ ok, uint8_t* __restrict__ result,
                                                                   uint32_t empty) {
    __shared__ uint32_t lds[54 * ZKV_BLOCK];
    size_t i = ((size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x) >> 1;
    if (i >= n) return;
    const uint32_t par = threadIdx.x & 1u;
    uint8_t res = empty ? 1 : 0;                                 // k = 0: the empty product is 1
    if (ok[i] && !empty) {
        L9Ref acc = l9_ref(lds + threadIdx.x);
        res = final_exp_prog_p(ws.f, ws.fe, ws.cap, (uint32_t)(8 * par * ws.cap + i) * 4u, acc) ? 1 : 0;
    }
    if (!par) result[i] = res;
}
// Pairs per call the one-loop kernel is built for (0: that many pairs take one loop each)
uint32_t pairing_group(uint32_t k) { return k < 2 ? 0u : k <= 2 ? 2u : k <= 4 ? 4u : k <= 8 ? 8u : 0u; }
void launch_pairing(size_t n, uint32_t k, const uint8_t* in, const Workspace& ws, uint8_t* result, uint8_t* ok, hipStream_t s) {
    if (!n) return;
    const unsigned grid = (unsigned)((2 * n + ZKV_BLOCK - 1) / ZKV_BLOCK);
    if (k == 0) (void)hipMemsetAsync(ok, 1, n, s);              // the empty product: valid input, result 1 (its F slot is set below)
    const uint32_t g = pairing_group(k);
    if (g && (size_t)k * n <= ws.cap) {                         // all pairs of a call resident: one Miller loop per call
        for (uint32_t j = 0; j < k; j++) hipLaunchKernelGGL(k_pairing_check, dim3(grid), dim3(ZKV_BLOCK), 0, s, n, k, j, (size_t)j * n, in, ws, ok);
        const dim3 hg((unsigned)((2 * n + PAIR_BLOCK - 1) / PAIR_BLOCK)), hb(PAIR_BLOCK);
        if (g == 2) hipLaunchKernelGGL(k_pairing_miller_g<2>, hg, hb, 0, s, n, k, ws, ok);
        else if (g == 4) hipLaunchKernelGGL(k_pairing_miller_g<4>, hg, hb, 0, s, n, k, ws, ok);
        else hipLaunchKernelGGL(k_pairing_miller_g<8>, hg, hb, 0, s, n, k, ws, ok);
    } else for (uint32_t j = 0; j < k; j++) {
        hipLaunchKernelGGL(k_pairing_check, dim3(grid), dim3(ZKV_BLOCK), 0, s, n, k, j, (size_t)0, in, ws, ok);
        hipLaunchKernelGGL(k_pairing_miller, dim3(grid), dim3(ZKV_BLOCK), 0, s, n, ws, ok);
    }
    hipLaunchKernelGGL(k_pairing_finalexp, dim3(grid), dim3(ZKV_BLOCK), 0, s, n, ws, ok, result, k == 0 ? 1u : 0u);
}

// Aggregate check (zkv_agg.h), Miller loop: G PROOFS PER LANE PAIR (2, 4 or 8) with one accumulator (miller_loop_pg) -- the squaring of f,
// a third of a step, is shared.  A block of 64 proofs belongs to L = 64 / G pairs: pair q takes proofs 64 b + q + p L, p < G (consecutive
// pairs read consecutive rows).  The running points go through the final exponentiation's scratch rows (ws.fe, unused at this stage).
// The group's Miller value times ML(alpha, beta) goes to the F slot of its first proof.  If a B fails the subgroup test the group has no
// value: the offender is rejected (g2bad 1) and the others marked for the ordinary kernels (g2bad 2).
template <int G>
__global__ __launch_bounds__(PAIR_BLOCK, 2) void k_agg_miller(size_t n, const VkTables* __restrict__ vk, Workspace ws, uint8_t* __restrict__ status) {
    __shared__ uint32_t lds[48 * PAIR_BLOCK];
    constexpr uint32_t L = 64u / G;
    const size_t gp = ((size_t)blockIdx.x * PAIR_BLOCK + threadIdx.x) >> 1;
    const size_t i0 = (gp / L) * 64 + (gp % L);
    if (i0 >= n) return;
    uint32_t alive = 0, mask = 0, abmask = 0;
#pragma unroll
    for (uint32_t p = 0; p < (uint32_t)G; p++) {
        const size_t i = i0 + p * L;
        const uint32_t f = i < n ? ws.flags[i] : 0u;
        if (f & FL_ALIVE) {
            alive |= 1u << p;
            if (!(f & FL_B_INF)) { mask |= 1u << p; if (!(f & FL_A_INF)) abmask |= 1u << p; }
        }
    }
    if (!alive) return;
    const uint32_t par = threadIdx.x & 1u;
    uint32_t* wl = lds + (threadIdx.x >> 6) * (48 * ZKV_BLOCK) + (threadIdx.x & 63u);
    LRef fm = l_ref(wl);
    SoaRef norm = {ws.norm, ws.cap, (uint32_t)i0 * 4u};
    SoaRef bsrc = {ws.prep + 32 * ws.cap, ws.cap, (uint32_t)(8 * par * ws.cap + i0) * 4u};
    SoaRW tq = {ws.fe, ws.cap, (uint32_t)(8 * par * ws.cap + i0) * 4u};
    const uint32_t fine = miller_loop_pg<G>(mask, abmask, norm, bsrc, tq, L * 4u, fm);
    if ((fine & mask) != mask) {
        if (!par) {
#pragma unroll 1
            for (uint32_t p = 0; p < (uint32_t)G; p++) {
                if (!((alive >> p) & 1u)) continue;
                const bool bad = ((mask & ~fine) >> p) & 1u;
                ws.g2bad[i0 + p * L] = bad ? 1u : 2u;
                if (bad) status[i0 + p * L] = ST_VERIFICATION_FAILED;
            }
        }
        return;
    }
    MRef ab = m_ref((uint32_t*)(vk->f_alpha_beta) + 8 * par, 1, 16);
    MRef out = m_ref(ws.f + (size_t)(8 * par) * ws.cap + i0, (uint32_t)ws.cap, 16);
    f12m_mul_body(out, fm, ab, false);
}
void launch_agg_miller(size_t n, uint32_t g, const VkTables* d_tab, const Workspace& ws, uint8_t* status, hipStream_t s) {
    if (!n) return;
    const size_t pairs = ((n + 63) / 64) * (64 / g);
    const dim3 grid((unsigned)((2 * pairs + PAIR_BLOCK - 1) / PAIR_BLOCK)), block(PAIR_BLOCK);
    if (g == 8) hipLaunchKernelGGL(k_agg_miller<8>, grid, block, 0, s, n, d_tab, ws, status);
    else if (g == 4) hipLaunchKernelGGL(k_agg_miller<4>, grid, block, 0, s, n, d_tab, ws, status);
    else hipLaunchKernelGGL(k_agg_miller<2>, grid, block, 0, s, n, d_tab, ws, status);
}
// Aggregate check: the product of the Miller values of a sub-batch, multiplied into the pseudo-proof's slot between its Miller loop and
// its final exponentiation.  One sub-batch per lane pair, the running product in LDS.  g proofs share a Miller value (1: k_miller2, one per
// proof; 2, 4, 8: k_agg_miller, kept at the group's first proof): sub-batch sb covers sub / g consecutive groups of its 64-proof block.
__global__ __launch_bounds__(ZKV_BLOCK, 2) void k_agg_fprod(size_t n, size_t n2, uint32_t sub, uint32_t g, Workspace ws, const uint32_t* __restrict__ agg, Workspace ws2) {
    __shared__ uint32_t lds[48 * ZKV_BLOCK];
    const size_t sb = ((size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x) >> 1;
    if (sb >= n2) return;
    if (!(ws2.flags[sb] & FL_ALIVE)) return;
    const uint32_t par = threadIdx.x & 1u;
    LRef acc = l_ref(lds + threadIdx.x);
    MRef P2 = m_ref(ws2.f + (size_t)(8 * par) * ws2.cap + sb, (uint32_t)ws2.cap, 16);
    f12m_copy(acc, P2);
    // sub <= 64: sub / g consecutive groups of one 64-proof block; sub = 128, 256: all 64 / g groups of 2 / 4 consecutive blocks
    const uint32_t L = 64u / g, w = (sub < 64u ? sub : 64u) / g, nblk = sub > 64u ? sub / 64u : 1u, per = sub > 64u ? 1u : 64u / sub;
#pragma unroll 1
    for (uint32_t b = 0; b < nblk; b++) {
        const size_t i0 = sub > 64u ? sb * sub + (size_t)b * 64 : (sb / per) * 64 + (sb % per) * w;
#pragma unroll 1
        for (uint32_t k = 0; k < w; k++) {
            bool in = false;                                    // both lanes of the pair read the same words
#pragma unroll 1
            for (uint32_t p = 0; p < g && !in; p++) {
                const size_t i = i0 + k + p * L;
                if (i < n) in = (agg[(size_t)AGG_W_FLAGS * ws.cap + i] & FL_ALIVE) && !ws.g2bad[i];
            }
            if (!in) continue;
            MRef Pi = m_ref(ws.f + (size_t)(8 * par) * ws.cap + i0 + k, (uint32_t)ws.cap, 16);
            f12m_mul(acc, acc, Pi);
        }
    }
    f12m_copy(P2, acc);
}
void launch_agg_fprod(size_t n, size_t n2, uint32_t sub, uint32_t g, const Workspace& ws, const uint32_t* agg, const Workspace& ws2, hipStream_t s) {
    if (!n2) return;
    hipLaunchKernelGGL(k_agg_fprod, dim3((unsigned)((2 * n2 + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n, n2, sub, g, ws, agg, ws2);
}

static inline unsigned pair_grid(size_t n) { return (unsigned)((2 * n + ZKV_BLOCK - 1) / ZKV_BLOCK); }
static inline unsigned hot_grid(size_t n) { return (unsigned)((2 * n + PAIR_BLOCK - 1) / PAIR_BLOCK); }

void launch_g2chk2(size_t n, const Workspace& ws, uint8_t* status, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_g2chk2, dim3(pair_grid(n)), dim3(ZKV_BLOCK), 0, s, n, ws, status);
}
void launch_miller2(size_t n, const VkTables* d_tab, const Workspace& ws, uint8_t* status, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_miller2, dim3(hot_grid(n)), dim3(PAIR_BLOCK), 0, s, n, d_tab, ws, status);
}
void launch_finalexp2(size_t n, const Workspace& ws, uint8_t* status, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_finalexp2, dim3(hot_grid(n)), dim3(PAIR_BLOCK), 0, s, n, ws, status);
}

}  // namespace zkv
