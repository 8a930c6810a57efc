// Aggregate check (zkv_agg.h), the one-proof-per-lane kernels: the per-proof G1 stage (vk_x, the coefficient, r A, r vk_x, r C), the
// per-sub-batch reduction (a wavefront = 64 proofs = one, two or four sub-batches: butterfly sums of the points and of the coefficients, the
// pseudo-proof's rows), and the kernel that turns the sub-batches' verdicts into statuses and re-arms the proofs of a failed
// sub-batch for the ordinary kernels.  The product of the proofs' Miller values is a lane-pair kernel (k_agg_fprod, k_pair.hip).
#include "zkv_internal.h"
#include "zkv_agg.h"

namespace zkv {

__global__ __launch_bounds__(64) void k_setup_agg(const VkRaw* __restrict__ raw, const VkTables* __restrict__ tab, AggTables* __restrict__ t) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j < AGG_ALPHA_POW) setup_agg_alpha(*raw, *t, j);
    if (j == AGG_ALPHA_POW) {
        for (int k = 0; k < 4; k++) t->beta[k] = fp_from_raw(raw->beta[k]);
        t->ok = (tab->vk_valid && !raw_g1_is_inf(raw->alpha) && !raw_g2_is_inf(raw->beta)) ? 1u : 0u;
    }
    const int w = j - AGG_ALPHA_POW - 1;                        // the window rows of the vk_x constant term (k_setup_base ran before)
    if (w >= 0 && w < MSM_MAX_WINDOWS) setup_agg_base_row(*tab, *t, w);
}
void launch_setup_agg(const VkRaw* d_raw, const VkTables* d_tab, AggTables* d_agg, hipStream_t s) {
    hipLaunchKernelGGL(k_setup_agg, dim3(2), dim3(64), 0, s, d_raw, d_tab, d_agg);     // 72 + 1 + 32 lanes at work
}

__device__ __forceinline__ void agg_st_g1j(uint32_t* agg, size_t cap, int word0, size_t i, const G1J& p) {
    ws_st(agg, cap, word0, i, p.x); ws_st(agg, cap, word0 + 8, i, p.y); ws_st(agg, cap, word0 + 16, i, p.z);
}
__device__ __forceinline__ G1J agg_ld_g1j(const uint32_t* agg, size_t cap, int word0, size_t i) {
    G1J p; p.x = ws_ld(agg, cap, word0, i); p.y = ws_ld(agg, cap, word0 + 8, i); p.z = ws_ld(agg, cap, word0 + 16, i);
    return p;
}

// Per proof: the coefficient r, A' <- r A' (normalised into the rows the Miller loop reads) and W = r C (Jacobian, summed per sub-batch
// later); for vk_x either the scalars r, r s_0, r s_1 mod r (sums != 0: one key, at most two per-proof signals) or U = r vk_x with vk_x as
// in k_msm.  The proof's flags word is saved and replaced by one that tells k_miller2 to skip both fixed pairs: its Miller value is then
// f_i = ML(r A', B) * ML(alpha, beta).
__device__ __forceinline__ void agg_st_fr(uint32_t* agg, size_t cap, int word0, size_t i, const Fr& a) {
#pragma unroll
    for (int k = 0; k < 8; k++) agg[(size_t)(word0 + k) * cap + i] = a.v[k];
}
__global__ __launch_bounds__(ZKV_BLOCK) void k_agg_g1(size_t n, const VkTables* __restrict__ vk, const InstTab* __restrict__ inst_tab, Workspace ws,
                                                      uint32_t* __restrict__ agg, AggSeed seed, uint32_t sums) {
    size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t flags0 = ws.flags[i];
    agg[(size_t)AGG_W_FLAGS * ws.cap + i] = flags0;
    if (!(flags0 & FL_ALIVE)) return;
    PrepOut in;
    in.ax = ws_ld(ws.prep, ws.cap, 0, i); in.ay = ws_ld(ws.prep, ws.cap, 8, i);
    in.cx = ws_ld(ws.prep, ws.cap, 16, i); in.cy = ws_ld(ws.prep, ws.cap, 24, i);
#pragma unroll 1
    for (uint32_t b = 0; b < vk->n_var; b++) {
#pragma unroll
        for (int k = 0; k < 8; k++) in.s[b][k] = ws.prep[(size_t)(64 + 8 * b + k) * ws.cap + i];
    }
    uint64_t r1, r2;
    agg_coeff(seed, (uint32_t)i, r1, r2);
    agg[(size_t)(AGG_W_R + 0) * ws.cap + i] = (uint32_t)r1; agg[(size_t)(AGG_W_R + 1) * ws.cap + i] = (uint32_t)(r1 >> 32);
    agg[(size_t)(AGG_W_R + 2) * ws.cap + i] = (uint32_t)r2; agg[(size_t)(AGG_W_R + 3) * ws.cap + i] = (uint32_t)(r2 >> 32);
    uint32_t flags = flags0;
    if (sums) {
        const Fr rm = agg_coeff_fr(r1, r2);
        agg_st_fr(agg, ws.cap, AGG_W_U, i, rm);
#pragma unroll 1
        for (uint32_t b = 0; b < (uint32_t)AGG_SUM_VARS; b++)
            agg_st_fr(agg, ws.cap, AGG_W_U + 8 + 8 * (int)b, i, b < vk->n_var ? fr_mul(fr_from_raw(in.s[b]), rm) : fr_zero());     // signals are < r (PREP)
    } else {
        G1J L;
        if (inst_tab) { const InstTab& t = inst_tab[flags >> 8]; L = msm_accumulate(*vk, in, t.base, t.base_inf); }
        else L = msm_accumulate(*vk, in);
        G1A la; uint32_t linf;
        g1j_to_affine(L, la, linf);
        agg_st_g1j(agg, ws.cap, AGG_W_U, i, linf ? g1j_infinity() : agg_mul(la.x, la.y, r1, r2));
    }
    if (inst_tab) flags &= 0xFFu;
    agg_st_g1j(agg, ws.cap, AGG_W_W, i, (flags & FL_C_INF) ? g1j_infinity() : agg_mul(in.cx, in.cy, r1, r2));
    flags |= FL_L_INF | FL_C_INF;                               // for the Miller kernel only: the fixed pairs are taken per sub-batch
    if (!(flags & FL_A_INF)) {
        const G1J a = agg_mul(in.ax, in.ay, r1, r2);
        if (fp_is_zero(a.z)) flags |= FL_A_INF;                 // r = 0 (probability 2^-128): the pair contributes 1
        else {
            const Fp iy = fp_inv(a.y);                          // no point of G1 has y = 0
            ws_st(ws.norm, ws.cap, 0, i, fp_mul(fp_mul(a.x, a.z), iy));                 // (X / Z^2) / (Y / Z^3)
            ws_st(ws.norm, ws.cap, 8, i, fp_mul(fp_mul(fp_sqr(a.z), a.z), iy));         // Z^3 / Y
        }
    }
    ws.flags[i] = flags;
}

__device__ __forceinline__ G1J g1j_xor(const G1J& p, int mask) {
    G1J r;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        r.x.v[k] = (uint32_t)__shfl_xor((int)p.x.v[k], mask, 64);
        r.y.v[k] = (uint32_t)__shfl_xor((int)p.y.v[k], mask, 64);
        r.z.v[k] = (uint32_t)__shfl_xor((int)p.z.v[k], mask, 64);
    }
    return r;
}
// One wavefront per 64 proofs = 64 / sub sub-batches (sub = 16, 32 or 64; larger sub-batches are parked per block for k_agg_combine).  Lane l holds proof 64 blockIdx + l: its U, W and coefficient
// words if the proof is still in the check (alive after PREP, B in the subgroup), nothing otherwise.  Butterflies inside each group of `sub`
// lanes give every lane its sub-batch's sums; E = (S1 - cnt - 1) alpha + S2 phi(alpha) comes from the lanes' table look-ups (bits lane,
// lane + sub, ... of S1 and S2) and a third butterfly -- every proof's Miller value and the pseudo-proof's carry one factor
// ML(alpha, beta), hence the cnt + 1.  The group's first lane writes the pseudo-proof (A := E, B := beta, vk_x := sum U, C := sum W) into
// the second workspace.
__device__ __forceinline__ Fr fr_xor(const Fr& a, int mask) {
    Fr r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = (uint32_t)__shfl_xor((int)a.v[k], mask, 64);
    return r;
}
__device__ __forceinline__ Fr agg_ld_fr(const uint32_t* agg, size_t cap, int word0, size_t i) {
    Fr r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = agg[(size_t)(word0 + k) * cap + i];
    return r;
}
// g proofs share a Miller value (k_agg_miller: members l, l + L, l + 2L, ... of the block, L = 64 / g): a sub-batch is then sub / g
// consecutive lanes of the lowest L and their partners -- the butterfly exchanges at distances 32 ... L and sub / (2 g) ... 1 -- and the
// number of ML(alpha, beta) factors to balance is the number of GROUPS with a proof in the check.  g = 1: sub consecutive lanes.
__global__ __launch_bounds__(ZKV_BLOCK) void k_agg_reduce(size_t n, uint32_t sub, uint32_t sums, uint32_t g, const VkTables* __restrict__ vk, Workspace ws,
                                                          const uint32_t* __restrict__ agg, const AggTables* __restrict__ tab, Workspace ws2,
                                                          uint8_t* __restrict__ status2, uint32_t park) {
    const size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    const uint32_t L = 64u / g, w = sub / g;
    const uint32_t dist = (63u & ~(L - 1u)) | (w - 1u);         // the exchange distances, one bit each: 32 ... L and w / 2 ... 1
    const uint32_t lane = (threadIdx.x & (w - 1u)) + (threadIdx.x / L) * w;
    const size_t sb = (size_t)blockIdx.x * (64u / sub) + (threadIdx.x & (L - 1u)) / w;
    bool in = false;
    if (i < n) in = (agg[(size_t)AGG_W_FLAGS * ws.cap + i] & FL_ALIVE) && !ws.g2bad[i];
    G1J U = g1j_infinity(), W = g1j_infinity();
    Fr Rm = fr_zero(), T0 = fr_zero(), T1 = fr_zero();         // sums of r, r s_0, r s_1 mod r (the scalar form of vk_x)
    uint32_t s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0}, cnt = in ? 1u : 0u;
    {                                                           // count a group once: at its first member in the check
        const unsigned long long inb = __ballot(in);
        unsigned long long lower = 0;
        for (uint32_t l = threadIdx.x & (L - 1u); l < threadIdx.x; l += L) lower |= 1ull << l;
        if (inb & lower) cnt = 0;
    }
    if (in) {
        if (sums) { Rm = agg_ld_fr(agg, ws.cap, AGG_W_U, i); T0 = agg_ld_fr(agg, ws.cap, AGG_W_U + 8, i); T1 = agg_ld_fr(agg, ws.cap, AGG_W_U + 16, i); }
        else U = agg_ld_g1j(agg, ws.cap, AGG_W_U, i);
        W = agg_ld_g1j(agg, ws.cap, AGG_W_W, i);
        s1[0] = agg[(size_t)(AGG_W_R + 0) * ws.cap + i]; s1[1] = agg[(size_t)(AGG_W_R + 1) * ws.cap + i];
        s2[0] = agg[(size_t)(AGG_W_R + 2) * ws.cap + i]; s2[1] = agg[(size_t)(AGG_W_R + 3) * ws.cap + i];
    }
#pragma unroll 1
    for (int m = 32; m >= 1; m >>= 1) {
        if (!(dist & (uint32_t)m)) continue;
        if (sums) { Rm = fr_add(Rm, fr_xor(Rm, m)); T0 = fr_add(T0, fr_xor(T0, m)); T1 = fr_add(T1, fr_xor(T1, m)); }
        else U = g1j_add(U, g1j_xor(U, m));
        W = g1j_add(W, g1j_xor(W, m));
        uint32_t c = 0;
        s1[0] = addc(s1[0], (uint32_t)__shfl_xor((int)s1[0], m, 64), c); s1[1] = addc(s1[1], (uint32_t)__shfl_xor((int)s1[1], m, 64), c);
        s1[2] = addc(s1[2], (uint32_t)__shfl_xor((int)s1[2], m, 64), c);
        c = 0;
        s2[0] = addc(s2[0], (uint32_t)__shfl_xor((int)s2[0], m, 64), c); s2[1] = addc(s2[1], (uint32_t)__shfl_xor((int)s2[1], m, 64), c);
        s2[2] = addc(s2[2], (uint32_t)__shfl_xor((int)s2[2], m, 64), c);
        cnt += (uint32_t)__shfl_xor((int)cnt, m, 64);
    }
    if (sums) {                                                 // every lane holds the sums: its windows of them, then a butterfly as for U itself
        uint32_t R[8], T[AGG_SUM_VARS][8];
        fr_to_raw(R, Rm); fr_to_raw(T[0], T0); fr_to_raw(T[1], T1);
        U = agg_u_share(*vk, *tab, lane, sub, R, T);
#pragma unroll 1
        for (int m = 32; m >= 1; m >>= 1) if (dist & (uint32_t)m) U = g1j_add(U, g1j_xor(U, m));
    }
    G1J E = g1j_infinity();
    if (tab) {                                                  // (a PLONK context has no (alpha, beta) pair: no table, no E)
        E = agg_e_share(*tab, lane, sub, ((uint64_t)s1[1] << 32) | s1[0], s1[2], ((uint64_t)s2[1] << 32) | s2[0], s2[2], cnt + 1u);
#pragma unroll 1
        for (int m = 32; m >= 1; m >>= 1) if (dist & (uint32_t)m) E = g1j_add(E, g1j_xor(E, m));
    }
    if (lane != 0) return;                                      // (a sub-batch past the end of the chunk has cnt = 0)
    if (park) {                                                 // sub-batches of 128 / 256 proofs: the 64-proof sums wait (Jacobian) for k_agg_combine
        uint32_t* row = ws2.fe;                                 // the pseudo-proofs' final-exponentiation scratch, unused until then
        ws_st(row, ws2.cap, 0, sb, E.x); ws_st(row, ws2.cap, 8, sb, E.y); ws_st(row, ws2.cap, 16, sb, E.z);
        ws_st(row, ws2.cap, 24, sb, U.x); ws_st(row, ws2.cap, 32, sb, U.y); ws_st(row, ws2.cap, 40, sb, U.z);
        ws_st(row, ws2.cap, 48, sb, W.x); ws_st(row, ws2.cap, 56, sb, W.y); ws_st(row, ws2.cap, 64, sb, W.z);
        row[(size_t)72 * ws2.cap + sb] = cnt;
        return;
    }
    ws2.g2bad[sb] = 0;
    if (cnt == 0) { ws2.flags[sb] = 0; status2[sb] = ST_OK; return; }       // nothing left to check in this sub-batch
    uint32_t flags = FL_ALIVE;
    G1Norm o;
    agg_normalize3(E, U, W, flags, o);
    ws_st(ws2.norm, ws2.cap, 0, sb, o.axs); ws_st(ws2.norm, ws2.cap, 8, sb, o.ays);
    ws_st(ws2.norm, ws2.cap, 16, sb, o.lxs); ws_st(ws2.norm, ws2.cap, 24, sb, o.lys);
    ws_st(ws2.norm, ws2.cap, 32, sb, o.cxs); ws_st(ws2.norm, ws2.cap, 40, sb, o.cys);
    if (tab) {
#pragma unroll 1
        for (int k = 0; k < 4; k++) ws_st(ws2.prep, ws2.cap, 32 + 8 * k, sb, tab->beta[k]);
    } else flags |= FL_B_INF;
    ws2.flags[sb] = flags;
    status2[sb] = ST_VERIFICATION_FAILED;
}

// Sub-batches of 128 / 256 proofs: the sums of `wide` = 2 / 4 consecutive 64-proof blocks (parked by k_agg_reduce) are added up by one lane
// and become ONE pseudo-proof.  Every block's E carries a "- 1" for a pseudo-proof's ML(alpha, beta) of its own, but only one pseudo-proof
// exists: the others' alpha are added back.
__global__ __launch_bounds__(ZKV_BLOCK) void k_agg_combine(size_t n64, size_t n2, uint32_t wide, const AggTables* __restrict__ tab, Workspace ws2,
                                                           uint8_t* __restrict__ status2) {
    const size_t j = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (j >= n2) return;
    const uint32_t* row = ws2.fe;
    G1J E = g1j_infinity(), U = g1j_infinity(), W = g1j_infinity();
    uint32_t cnt = 0, blocks = 0;
#pragma unroll 1
    for (uint32_t t = 0; t < wide; t++) {
        const size_t sb = j * wide + t;
        if (sb >= n64) break;
        const uint32_t c = row[(size_t)72 * ws2.cap + sb];
        if (!c) continue;
        cnt += c; blocks++;
        G1J p;
        p.x = ws_ld(row, ws2.cap, 0, sb); p.y = ws_ld(row, ws2.cap, 8, sb); p.z = ws_ld(row, ws2.cap, 16, sb); E = g1j_add(E, p);
        p.x = ws_ld(row, ws2.cap, 24, sb); p.y = ws_ld(row, ws2.cap, 32, sb); p.z = ws_ld(row, ws2.cap, 40, sb); U = g1j_add(U, p);
        p.x = ws_ld(row, ws2.cap, 48, sb); p.y = ws_ld(row, ws2.cap, 56, sb); p.z = ws_ld(row, ws2.cap, 64, sb); W = g1j_add(W, p);
    }
    ws2.g2bad[j] = 0;
    if (cnt == 0) { ws2.flags[j] = 0; status2[j] = ST_OK; return; }
    if (tab && blocks > 1u) {
        const uint32_t extra = blocks - 1u;                     // 1 .. 3
        if (extra & 1u) { const G1A a = tab->alpha_pow[0]; E = g1j_add_affine(E, a.x, a.y); }
        if (extra & 2u) { const G1A a = tab->alpha_pow[1]; E = g1j_add_affine(E, a.x, a.y); }
    }
    uint32_t flags = FL_ALIVE;
    G1Norm o;
    agg_normalize3(E, U, W, flags, o);
    ws_st(ws2.norm, ws2.cap, 0, j, o.axs); ws_st(ws2.norm, ws2.cap, 8, j, o.ays);
    ws_st(ws2.norm, ws2.cap, 16, j, o.lxs); ws_st(ws2.norm, ws2.cap, 24, j, o.lys);
    ws_st(ws2.norm, ws2.cap, 32, j, o.cxs); ws_st(ws2.norm, ws2.cap, 40, j, o.cys);
    if (tab) {
#pragma unroll 1
        for (int k = 0; k < 4; k++) ws_st(ws2.prep, ws2.cap, 32 + 8 * k, j, tab->beta[k]);
    } else flags |= FL_B_INF;
    ws2.flags[j] = flags;
    status2[j] = ST_VERIFICATION_FAILED;
}
void launch_agg_combine(size_t n64, size_t n2, uint32_t wide, const AggTables* tab, const Workspace& ws2, uint8_t* status2, hipStream_t s) {
    if (!n2) return;
    hipLaunchKernelGGL(k_agg_combine, dim3((unsigned)((n2 + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n64, n2, wide, tab, ws2, status2);
}

// Verdicts: a proof that was in a sub-batch whose check passed is accepted; one in a failed sub-batch is queued for the ordinary
// kernels: its index goes into a dense list (slots reserved per wavefront with one atomic add -- the order does not matter), so that
// those kernels run on full wavefronts spread over the whole chip however the failures are placed in the batch (with the proofs left
// where they were, a batch whose rejects sit at a fixed position of every 64 used half of the XCDs: 28.7 instead of 14 ms).
// Proofs PREP rejected or whose B failed the subgroup test already have their final status.  counters: [0] sub-batches checked,
// [1] sub-batches that failed, [2] length of the list (reset per chunk).
__global__ __launch_bounds__(ZKV_BLOCK) void k_agg_mark(size_t n, uint32_t sub, uint32_t g, Workspace ws, const uint32_t* __restrict__ agg,
                                                        const uint8_t* __restrict__ status2, uint8_t* __restrict__ status, unsigned long long* __restrict__ counters,
                                                        uint32_t* __restrict__ idx) {
    const size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    bool again = false;
    if (i < n) {
        const uint32_t L = 64u / g, w = (sub < 64u ? sub : 64u) / g;
        const size_t sb = sub > 64u ? i / sub : (size_t)blockIdx.x * (64u / sub) + (threadIdx.x & (L - 1u)) / w;      // as in k_agg_reduce / k_agg_combine
        const bool first = sub > 64u ? (i % sub) == 0 : (threadIdx.x < L && (threadIdx.x & (w - 1u)) == 0);
        const bool passed = status2[sb] == ST_OK;
        if (first) { atomicAdd(&counters[0], 1ull); if (!passed) atomicAdd(&counters[1], 1ull); }
        const uint32_t flags0 = agg[(size_t)AGG_W_FLAGS * ws.cap + i];
        const uint32_t bad = ws.g2bad[i];
        if (flags0 & FL_ALIVE) {
            if (bad == 2u) again = true;                        // another B of its group failed the subgroup test: the group had no Miller value
            else if (!bad) { if (passed) status[i] = ST_OK; else again = true; }
        }
    }
    const unsigned long long m = __ballot(again);
    if (!m) return;
    unsigned long long base = 0;
    if (threadIdx.x == 0) base = atomicAdd(&counters[2], (unsigned long long)__popcll(m));
    base = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(base >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)base, 0, 64);
    if (again) idx[base + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = (uint32_t)i;
}
// Slot j of the dense workspace ws3 receives the PREP rows and flags of proof idx[j]; slots past the end of the list are switched off.
// (ws3 shares its scratch rows -- norm, f, fe -- with ws: nothing of the aggregate pass is needed any more.)
__global__ __launch_bounds__(ZKV_BLOCK) void k_agg_gather(size_t n, Workspace ws, const uint32_t* __restrict__ agg, const unsigned long long* __restrict__ counters,
                                                          const uint32_t* __restrict__ idx, Workspace ws3, uint8_t* __restrict__ status3) {
    const size_t j = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (j >= n) return;
    if (j >= counters[2]) { ws3.flags[j] = 0; return; }
    const size_t i = idx[j];
#pragma unroll 4
    for (int k = 0; k < WS_PREP_WORDS; k++) ws3.prep[(size_t)k * ws3.cap + j] = ws.prep[(size_t)k * ws.cap + i];
    ws3.flags[j] = agg[(size_t)AGG_W_FLAGS * ws.cap + i];
    ws3.g2bad[j] = 0;
    status3[j] = ST_VERIFICATION_FAILED;
}
__global__ __launch_bounds__(ZKV_BLOCK) void k_agg_scatter(size_t n, const unsigned long long* __restrict__ counters, const uint32_t* __restrict__ idx,
                                                           const uint8_t* __restrict__ status3, uint8_t* __restrict__ status) {
    const size_t j = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (j < n && j < counters[2]) status[idx[j]] = status3[j];
}

// PLONK contexts (two FIXED pairs: e(D, [1]_2) e(-Q, [tau]_2) == 1): the aggregate check needs no per-proof Miller loop at all --
// prod_i (e(D_i, [1]_2) e(-Q_i, [tau]_2))^{r_i} = e(sum r_i D_i, [1]_2) e(sum r_i (-Q_i), [tau]_2).  k_plonk_prep runs unchanged; this
// kernel turns its normalised rows (x/y, 1/y of D and -Q) back into affine points (one inversion), scales them by r_i and parks them
// where the Groth16 form keeps U and W.  A zero coefficient (probability 2^-128) is replaced by 1: the second pass re-checks a proof
// through its SCALED points, which is the proof's own check raised to r_i.
__global__ __launch_bounds__(ZKV_BLOCK) void k_agg_plonk_g1(size_t n, Workspace ws, uint32_t* __restrict__ agg, AggSeed seed) {
    size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t flags = ws.flags[i];
    agg[(size_t)AGG_W_FLAGS * ws.cap + i] = flags;
    if (!(flags & FL_ALIVE)) return;
    uint64_t r1, r2;
    agg_coeff(seed, (uint32_t)i, r1, r2);
    if (!(r1 | r2)) r1 = 1;
    const bool dinf = (flags & FL_L_INF) != 0, qinf = (flags & FL_C_INF) != 0;
    const Fp one = fp_one();
    const Fp dys = dinf ? one : ws_ld(ws.norm, ws.cap, 24, i), qys = qinf ? one : ws_ld(ws.norm, ws.cap, 40, i);      // 1 / y
    const Fp inv = fp_inv(fp_mul(dys, qys));
    const Fp dy = fp_mul(inv, qys), qy = fp_mul(inv, dys);
    agg_st_g1j(agg, ws.cap, AGG_W_U, i, dinf ? g1j_infinity() : agg_mul(fp_mul(ws_ld(ws.norm, ws.cap, 16, i), dy), dy, r1, r2));
    agg_st_g1j(agg, ws.cap, AGG_W_W, i, qinf ? g1j_infinity() : agg_mul(fp_mul(ws_ld(ws.norm, ws.cap, 32, i), qy), qy, r1, r2));
}
// ... and for the second pass: the scaled points of the listed proofs, normalised into the dense workspace's rows
__global__ __launch_bounds__(ZKV_BLOCK) void k_agg_plonk_norm(size_t n, Workspace ws, const uint32_t* __restrict__ agg, const unsigned long long* __restrict__ counters,
                                                              const uint32_t* __restrict__ idx, Workspace ws3, uint8_t* __restrict__ status3) {
    const size_t j = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (j >= n) return;
    if (j >= counters[2]) { ws3.flags[j] = 0; return; }
    const size_t i = idx[j];
    uint32_t flags = FL_ALIVE | FL_B_INF;
    G1Norm o;
    agg_normalize3(g1j_infinity(), agg_ld_g1j(agg, ws.cap, AGG_W_U, i), agg_ld_g1j(agg, ws.cap, AGG_W_W, i), flags, o);
    ws_st(ws3.norm, ws3.cap, 0, j, o.axs); ws_st(ws3.norm, ws3.cap, 8, j, o.ays);
    ws_st(ws3.norm, ws3.cap, 16, j, o.lxs); ws_st(ws3.norm, ws3.cap, 24, j, o.lys);
    ws_st(ws3.norm, ws3.cap, 32, j, o.cxs); ws_st(ws3.norm, ws3.cap, 40, j, o.cys);
    ws3.flags[j] = flags;
    ws3.g2bad[j] = 0;
    status3[j] = ST_VERIFICATION_FAILED;
}
void launch_agg_plonk_g1(size_t n, const Workspace& ws, uint32_t* agg, const AggSeed& seed, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_agg_plonk_g1, dim3((unsigned)((n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n, ws, agg, seed);
}
void launch_agg_plonk_norm(size_t n, const Workspace& ws, const uint32_t* agg, const unsigned long long* counters, const uint32_t* idx, const Workspace& ws3,
                           uint8_t* status3, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_agg_plonk_norm, dim3((unsigned)((n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n, ws, agg, counters, idx, ws3, status3);
}

void launch_agg_g1(size_t n, const VkTables* d_tab, const InstTab* inst_tab, const Workspace& ws, uint32_t* agg, const AggSeed& seed, bool sums, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_agg_g1, dim3((unsigned)((n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n, d_tab, inst_tab, ws, agg, seed, sums ? 1u : 0u);
}
void launch_agg_reduce(size_t n, uint32_t sub, bool sums, uint32_t g, const VkTables* d_tab, const Workspace& ws, const uint32_t* agg, const AggTables* tab,
                       const Workspace& ws2, uint8_t* status2, bool park, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_agg_reduce, dim3((unsigned)((n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n, sub, sums ? 1u : 0u, g, d_tab, ws, agg,
                       tab, ws2, status2, park ? 1u : 0u);
}
void launch_agg_mark(size_t n, uint32_t sub, uint32_t g, const Workspace& ws, const uint32_t* agg, const uint8_t* status2, uint8_t* status, unsigned long long* counters,
                     uint32_t* idx, hipStream_t s) {
    if (!n) return;
    (void)hipMemsetAsync(counters + 2, 0, sizeof(unsigned long long), s);
    hipLaunchKernelGGL(k_agg_mark, dim3((unsigned)((n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n, sub, g, ws, agg, status2, status, counters, idx);
}
void launch_agg_gather(size_t n, const Workspace& ws, const uint32_t* agg, const unsigned long long* counters, const uint32_t* idx, const Workspace& ws3,
                       uint8_t* status3, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_agg_gather, dim3((unsigned)((n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n, ws, agg, counters, idx, ws3, status3);
}
void launch_agg_scatter(size_t n, const unsigned long long* counters, const uint32_t* idx, const uint8_t* status3, uint8_t* status, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_agg_scatter, dim3((unsigned)((n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, n, counters, idx, status3, status);
}

}  // namespace zkv
