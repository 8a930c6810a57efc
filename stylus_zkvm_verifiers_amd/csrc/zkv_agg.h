// Aggregate check of a sub-batch of Groth16 proofs (opt-in: zkv_ctx_set_aggregate_check).
//
// The reference verifies one proof per call: e(-A, B) e(alpha, beta) e(vk_x, gamma) e(C, delta) == 1 (common/groth16.rs:60-72,
// 109-128).  A batch service may instead check a sub-batch of 16 ... 256 proofs with ONE final exponentiation: for coefficients r_i unknown to
// whoever produced the proofs,
//     prod_i e(r_i (-A_i), B_i)  *  e(sum_i r_i vk_x_i, gamma)  *  e(sum_i r_i C_i, delta)  *  e((sum_i r_i) alpha, beta)  ==  1
// holds when every proof of the sub-batch verifies, and with probability 2^-128 over the coefficients otherwise.  Everything before
// the pairing check stays per proof and deterministic (seal parsing, selector, range checks, curve and subgroup membership of A, B,
// C): only the verdict of the pairing equation is shared.  A sub-batch whose aggregate check fails is verified again proof by proof
// with the ordinary kernels, so the statuses a caller sees are the deterministic ones (up to that 2^-128).
//
// Coefficients: r_i = r1 + r2 lambda (mod r) with r1, r2 the first two 64-bit words of SHA-256(seed || call counter || index); the map
// (r1, r2) -> r_i is injective on [0, 2^64)^2 (the GLV lattice of BN254 has no non-zero vector that short), so r_i is uniform over
// 2^128 values, and r_i P = r1 P + r2 phi(P) costs 64 doublings and at most 64 additions (phi(P) = (beta x, y), P + phi(P) = -phi^2(P)).
//
// vk_x: sum_i r_i vk_x_i = (sum_i r_i) base + sum_b (sum_i r_i s_ib) IC_b  with base = IC_0 + the key's fixed signals: for one key and at
// most two per-proof signals the proofs only contribute the scalars r_i and r_i s_ib (mod r); a sub-batch's lanes look the sums up in
// the key's 8-bit window tables (one window per lane) and a butterfly adds the shares -- no per-proof vk_x, no r_i vk_x_i.  Verifier
// sets (a base per proof) and keys with more signals keep the per-proof form U_i = r_i vk_x_i.
#pragma once
#include "zkv_verify.h"
#if !defined(ZKV_PAIRED)
#include "zkv_scalar.h"     // the scalar field Fr
#endif

namespace zkv {

constexpr int AGG_ALPHA_POW = 72;        // 2^j alpha for the bits of sum r1, sum r2 (at most 64 summands of 64 bits: 70 bits)
constexpr int WS_AGG_WORDS = 56;         // per proof: U = r vk_x (24), or r | r s_0 | r s_1 mod r (3 x 8) | W = r C (24) | r1, r2 (4) | the flags word PREP left (1) | pad (3)
constexpr int AGG_W_U = 0, AGG_W_W = 24, AGG_W_R = 48, AGG_W_FLAGS = 52;
constexpr int AGG_SUM_VARS = 2;          // the scalar-sum form of vk_x needs 8 (1 + n_var) <= 24 words per proof

struct AggTables {
    G1A alpha_pow[AGG_ALPHA_POW];        // 2^j alpha, affine
    Fp beta[4];                          // beta as the PREP rows hold a proof's B: x.c0 x.c1 y.c0 y.c1
    uint32_t ok;                         // the key supports the aggregate check (valid, alpha and beta finite)
    uint32_t pad[7];
    G1A base_win[MSM_MAX_WINDOWS][MSM_DIGITS];     // d * 256^w * base (the vk_x constant term), like VkTables::msm: 512 KB
};
struct AggSeed { uint32_t w[8]; uint32_t call; };      // 32 secret bytes and the number of the chunk they are used on

#if !defined(ZKV_PAIRED)
ZKV_HD void agg_coeff(const AggSeed& seed, uint32_t index, uint64_t& r1, uint64_t& r2) {
    uint32_t h[8], w[16];
    sha256_init(h);
    for (int i = 0; i < 8; i++) w[i] = seed.w[i];
    w[8] = seed.call; w[9] = index; w[10] = 0x80000000u;       // 40 message bytes, then the padding
    for (int i = 11; i < 15; i++) w[i] = 0;
    w[15] = 40u * 8u;
    sha256_compress(h, w);
    r1 = ((uint64_t)h[0] << 32) | h[1]; r2 = ((uint64_t)h[2] << 32) | h[3];
}
// r1 P + r2 phi(P) for affine P != O: one joint bit per step, the three non-zero digit values from {P, phi(P), -phi^2(P)}
ZKV_HD_NI G1J agg_mul(const Fp& x, const Fp& y, uint64_t r1, uint64_t r2) {
    const Fp beta = ZKV_GLV_BETA;
    const Fp bx = fp_mul(x, beta), bbx = fp_mul(bx, beta), ny = fp_neg(y);
    G1J acc = g1j_infinity();
#pragma unroll 1
    for (int b = 63; b >= 0; b--) {
        acc = g1j_dbl(acc);
        const uint32_t d = (uint32_t)((r1 >> b) & 1u) | ((uint32_t)((r2 >> b) & 1u) << 1);
        if (d) {
            const Fp qx = d == 1 ? x : d == 2 ? bx : bbx;
            const Fp qy = d == 3 ? ny : y;
            acc = g1j_add_affine(acc, qx, qy);
        }
    }
    return acc;
}
// One lane's share of E = (S1 - c) alpha + S2 phi(alpha), `sub` lanes (16, 32 or 64) per sub-batch: bits lane, lane + sub, ... of S1
// and S2 (below AGG_ALPHA_POW), and bit `lane` of c (c <= sub + 1: seven bits) with the negated table entry.  The sum of the shares is E.
ZKV_HD G1J agg_e_share(const AggTables& t, uint32_t lane, uint32_t sub, uint64_t s1lo, uint32_t s1hi, uint64_t s2lo, uint32_t s2hi, uint32_t c) {
    const Fp beta = ZKV_GLV_BETA;
    G1J acc = g1j_infinity();
#pragma unroll 1
    for (uint32_t j = lane; j < (uint32_t)AGG_ALPHA_POW; j += sub) {
        const uint32_t b1 = j < 64u ? (uint32_t)(s1lo >> j) & 1u : (s1hi >> (j - 64u)) & 1u;
        const uint32_t b2 = j < 64u ? (uint32_t)(s2lo >> j) & 1u : (s2hi >> (j - 64u)) & 1u;
        const G1A e = t.alpha_pow[j];
        if (b1) acc = g1j_add_affine(acc, e.x, e.y);
        if (b2) acc = g1j_add_affine(acc, fp_mul(e.x, beta), e.y);
    }
    if (lane < 7u && ((c >> lane) & 1u)) { const G1A e = t.alpha_pow[lane]; acc = g1j_add_affine(acc, e.x, fp_neg(e.y)); }
    return acc;
}
// r = r1 + r2 lambda mod r, Montgomery form
ZKV_HD Fr agg_coeff_fr(uint64_t r1, uint64_t r2) {
    const uint32_t lam[8] = ZKV_GLV_LAMBDA;
    uint32_t a[8] = {(uint32_t)r1, (uint32_t)(r1 >> 32), 0, 0, 0, 0, 0, 0}, b[8] = {(uint32_t)r2, (uint32_t)(r2 >> 32), 0, 0, 0, 0, 0, 0};
    return fr_add(fr_from_raw(a), fr_mul(fr_from_raw(b), fr_from_raw(lam)));
}
// One lane's share of U = R base + sum_b T_b IC_b (R, T_b canonical, 8 limbs each): the 32 windows of R and of every T_b are numbered
// through; the lane adds the table entries of windows lane, lane + sub, ...
ZKV_HD G1J agg_u_share(const VkTables& vk, const AggTables& t, uint32_t lane, uint32_t sub, const uint32_t R[8], const uint32_t (*T)[8]) {
    G1J acc = g1j_infinity();
    const uint32_t total = (uint32_t)MSM_MAX_WINDOWS * (1u + vk.n_var);
#pragma unroll 1
    for (uint32_t g = lane; g < total; g += sub) {
        const uint32_t k = g / (uint32_t)MSM_MAX_WINDOWS, w = g % (uint32_t)MSM_MAX_WINDOWS;
        const uint32_t* sc = k == 0 ? R : T[k - 1];
        const uint32_t d = (sc[w >> 2] >> ((w & 3u) * 8u)) & 255u;
        if (!d) continue;
        if (k == 0) { if (vk.base_inf) continue; const G1A e = t.base_win[w][d]; acc = g1j_add_affine(acc, e.x, e.y); }
        else { if (!vk.var_windows[k - 1]) continue; const G1A e = vk.msm[k - 1][w][d]; acc = g1j_add_affine(acc, e.x, e.y); }      // var_windows 0: IC_b is infinity
    }
    return acc;
}
// x/y and 1/y of three Jacobian points with one inversion (the form the Miller loop evaluates lines at); an infinite point sets its
// flag and leaves zeros.
ZKV_HD void agg_normalize3(const G1J& e, const G1J& u, const G1J& w, uint32_t& flags, G1Norm& o) {
    const Fp one = fp_one();
    const bool ei = fp_is_zero(e.z), ui = fp_is_zero(u.z), wi = fp_is_zero(w.z);
    if (ei) flags |= FL_A_INF;
    if (ui) flags |= FL_L_INF;
    if (wi) flags |= FL_C_INF;
    const Fp ye = ei ? one : e.y, yu = ui ? one : u.y, yw = wi ? one : w.y;
    const Fp t = fp_mul(ye, yu);
    const Fp inv = fp_inv(fp_mul(t, yw));
    const Fp iyw = fp_mul(inv, t), v = fp_mul(inv, yw);
    const Fp iyu = fp_mul(v, ye), iye = fp_mul(v, yu);
    o.axs = fp_mul(fp_mul(e.x, e.z), iye); o.ays = fp_mul(fp_mul(fp_sqr(e.z), e.z), iye);
    o.lxs = fp_mul(fp_mul(u.x, u.z), iyu); o.lys = fp_mul(fp_mul(fp_sqr(u.z), u.z), iyu);
    o.cxs = fp_mul(fp_mul(w.x, w.z), iyw); o.cys = fp_mul(fp_mul(fp_sqr(w.z), w.z), iyw);
}
// 2^j alpha (affine) for one j
ZKV_HD void setup_agg_alpha(const VkRaw& vk, AggTables& t, int j) {
    if (raw_g1_is_inf(vk.alpha)) { t.alpha_pow[j].x = fp_zero(); t.alpha_pow[j].y = fp_zero(); return; }
    G1J p; p.x = fp_from_raw(vk.alpha[0]); p.y = fp_from_raw(vk.alpha[1]); p.z = fp_one();
#pragma unroll 1
    for (int i = 0; i < j; i++) p = g1j_dbl(p);
    uint32_t inf;
    g1j_to_affine(p, t.alpha_pow[j], inf);
}
ZKV_HD void setup_agg_base_row(const VkTables& vk, AggTables& t, int w) {
    if (!vk.base_inf) setup_window_row(vk.base.x, vk.base.y, w, t.base_win[w]);
}
#endif  // !ZKV_PAIRED

}  // namespace zkv
