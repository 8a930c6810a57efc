// The BN254 scalar field Fr and the GLV decomposition of G1 scalars: shared by the PLONK stage (zkv_plonk.h), the ecMul kernel
// (k_precompile.hip: g1_mul_glv) and the aggregate check (zkv_agg.h: coefficients and summed scalars).  One proof / call per lane.
#pragma once
#include "zkv_curve.h"

namespace zkv {

// ---------------------------------------------------------------- scalar field Fr: Montgomery form with R = 2^261, CANONICAL (< r)
// Stored as 8 x 32-bit limbs like Fp and multiplied on the same 9 x 29-bit column form (zkv_field.h: one v_mad_u64_u32 per term, no
// carry handling inside a column); round 2 ran an 8 x 32-bit CIOS loop here -- about 600 instructions per product against 300, 13 %
// of the PLONK stage.  Unlike Fp the results are kept canonical (one conditional subtraction after the reduction, whose output is
// below V / 2^261 + r < 2r for V < r^2): the transcripts hash canonical bytes and there are few additions to save.
struct Fr { uint32_t v[8]; };
ZKV_HD Fr fr_zero() { Fr r; for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
ZKV_HD Fr fr_one() { Fr r = ZKV_FR_ONE; return r; }
ZKV_HD bool fr_is_zero(const Fr& a) { uint32_t o = 0; for (int i = 0; i < 8; i++) o |= a.v[i]; return o == 0; }
ZKV_HD Fr fr_add(const Fr& a, const Fr& b) {
    const uint32_t M[8] = ZKV_FR_R_LIMBS;
    Fr t, s; uint32_t c = 0, br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t.v[i] = addc(a.v[i], b.v[i], c);          // < 2r < 2^255: no carry out
#pragma unroll
    for (int i = 0; i < 8; i++) s.v[i] = subb(t.v[i], M[i], br);
#pragma unroll
    for (int i = 0; i < 8; i++) t.v[i] = br ? t.v[i] : s.v[i];
    return t;
}
ZKV_HD Fr fr_sub(const Fr& a, const Fr& b) {
    const uint32_t M[8] = ZKV_FR_R_LIMBS;
    Fr t; uint32_t br = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t.v[i] = subb(a.v[i], b.v[i], br);
    const uint32_t mask = 0u - br;
#pragma unroll
    for (int i = 0; i < 8; i++) t.v[i] = addc(t.v[i], M[i] & mask, c);
    return t;
}
ZKV_HD Fr fr_neg(const Fr& a) { return fr_sub(fr_zero(), a); }
#if defined(ZKV_COUNT_FP_MUL)
static thread_local unsigned long long zkv_fr_mul_counter = 0;     // host-only op counter (tests/host_sim)
#endif
// a b 2^-261 mod r, canonical.  Operands: any values below 2^256 whose product stays below r 2^261 (canonical values, and the raw
// 256-bit integers fr_from_raw / fr_to_raw pass in).
ZKV_HD_NI Fr fr_mul(Fr a, Fr b) {
#if defined(ZKV_COUNT_FP_MUL)
    zkv_fr_mul_counter++;
#endif
    const uint32_t M[8] = ZKV_FR_R_LIMBS;
    const uint32_t R29[9] = ZKV_FR_R29_LIMBS;
    const uint32_t M29 = 0x1fffffffu;
    uint32_t x[9], y[9];
    { Fp t; for (int i = 0; i < 8; i++) t.v[i] = a.v[i]; fp_unpack29(t, x); for (int i = 0; i < 8; i++) t.v[i] = b.v[i]; fp_unpack29(t, y); }
    uint64_t col[18];
#pragma unroll
    for (int k = 0; k < 18; k++) col[k] = 0;
    fp_mac81(col, x, y);
    ZKV_COUNT_MADS(81);
#pragma unroll
    for (int i = 0; i < 9; i++) {                 // Montgomery reduction, one 29-bit digit per step (fp_reduce_cols with r's limbs)
        const uint32_t m = ((uint32_t)col[i] * ZKV_FR_INV29) & M29;
#pragma unroll
        for (int j = 0; j < 9; j++) col[i + j] += (uint64_t)m * R29[j];
        col[i + 1] += col[i] >> 29;
    }
    uint32_t r9[9];
#pragma unroll
    for (int k = 9; k < 17; k++) { r9[k - 9] = (uint32_t)col[k] & M29; col[k + 1] += col[k] >> 29; }
    r9[8] = (uint32_t)col[17];
    Fr o, s;
#pragma unroll
    for (int w = 0; w < 8; w++) {                 // pack 9 x 29 -> 8 x 32 (the value is < 2r < 2^255)
        const int bit = 32 * w, k = bit / 29, sh = bit - 29 * k, got = 29 - sh;
        uint32_t v = r9[k] >> sh;
        if (k + 1 < 9) v |= r9[k + 1] << got;
        if (got + 29 < 32 && k + 2 < 9) v |= r9[k + 2] << (got + 29);
        o.v[w] = v;
    }
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s.v[i] = subb(o.v[i], M[i], br);
#pragma unroll
    for (int i = 0; i < 8; i++) o.v[i] = br ? o.v[i] : s.v[i];
    return o;
}
ZKV_HD Fr fr_from_raw(const uint32_t* limbs) {               // canonical value < r -> Montgomery form
    Fr t, r2 = ZKV_FR_R2;
    for (int i = 0; i < 8; i++) t.v[i] = limbs[i];
    return fr_mul(t, r2);
}
ZKV_HD void fr_to_raw(uint32_t* limbs, const Fr& a) {
    Fr one = fr_zero(); one.v[0] = 1;
    Fr t = fr_mul(a, one);
    for (int i = 0; i < 8; i++) limbs[i] = t.v[i];
}
// any 256-bit value mod r (challenges are raw SHA-256 outputs): r > 2^253, so at most five subtractions
ZKV_HD Fr fr_from_raw_reduce(const uint32_t* limbs) {
    const uint32_t M[8] = ZKV_FR_R_LIMBS;
    uint32_t t[8];
    for (int i = 0; i < 8; i++) t[i] = limbs[i];
#pragma unroll 1
    for (int k = 0; k < 5; k++) {
        uint32_t s[8], br = 0;
        for (int i = 0; i < 8; i++) s[i] = subb(t[i], M[i], br);
        if (!br) for (int i = 0; i < 8; i++) t[i] = s[i];
    }
    return fr_from_raw(t);
}
ZKV_HD Fr fr_pow(const Fr& a, const uint32_t* e, int bits) {   // left to right
    Fr acc = fr_one();
#pragma unroll 1
    for (int i = bits - 1; i >= 0; i--) {
        acc = fr_mul(acc, acc);
        if ((e[i >> 5] >> (i & 31)) & 1u) acc = fr_mul(acc, a);
    }
    return acc;
}
ZKV_HD Fr fr_inv_fermat(const Fr& a) { const uint32_t E[8] = ZKV_FR_RM2_LIMBS; return fr_pow(a, E, 254); }     // inv(0) = 0; the check of fr_inv
// inv(0) = 0.  Division steps (zkv_modinv.h) on the Montgomery residue, then one multiplication by R^3.
ZKV_HD Fr fr_inv(const Fr& a) {
    const int32_t M[9] = ZKV_FR_M30_LIMBS;
    const Fr r3 = ZKV_FR_R3;
    Fr t = a;
    modinv30(t.v, M, ZKV_FR_MINV30);
    return fr_mul(t, r3);
}

// ---- GLV: k = k1 + k2 lambda (mod r) with |k1|, |k2| < 2^128, lambda P = phi(P) = (beta x, y).
// Babai rounding against the basis (a1, -n), (n, b2), n = 2u + 1, a1 = 6u^2 + 2u, b2 = 6u^2 + 4u + 1:  c1 = floor(k g1 / 2^256),
// c2 = floor(k g2 / 2^256) with g1 = floor(2^256 b2 / r), g2 = floor(2^256 n / r);  k1 = k - c1 a1 - c2 n,  k2 = c1 n - c2 b2.
// The identity k1 + k2 lambda = k holds for ANY c1, c2 (both basis vectors are 0 mod r); the truncations only cost magnitude:
// |k1|, |k2| <= 2^127 on 2 x 10^5 random and edge scalars in the model (gen_constants.py documents the constants).
template <int NA, int NB> ZKV_HD void glv_mul(const uint32_t (&a)[NA], const uint32_t (&b)[NB], uint32_t (&out)[NA + NB]) {
    for (int i = 0; i < NA + NB; i++) out[i] = 0;
#pragma unroll 1
    for (int i = 0; i < NA; i++) {
        uint64_t c = 0;
#pragma unroll 1
        for (int j = 0; j < NB; j++) { c += (uint64_t)a[i] * b[j] + out[i + j]; out[i + j] = (uint32_t)c; c >>= 32; }
        out[i + NB] = (uint32_t)c;
    }
}
// m <- |x - y - z| on six words (two's complement; the true value is below 2^129 in magnitude), returns 1 when negative
ZKV_HD uint32_t glv_diff(const uint32_t* x, const uint32_t* y, const uint32_t* z, int nz, uint32_t (&m)[5]) {
    uint32_t t[6]; uint32_t b = 0;
#pragma unroll 1
    for (int i = 0; i < 6; i++) t[i] = subb(x[i], y[i], b);
    if (z) { b = 0;
#pragma unroll 1
        for (int i = 0; i < 6; i++) t[i] = subb(t[i], i < nz ? z[i] : 0u, b); }
    const uint32_t neg = t[5] >> 31, mask = 0u - neg;
    uint32_t c = neg;
#pragma unroll 1
    for (int i = 0; i < 6; i++) { const uint64_t v = (uint64_t)(t[i] ^ mask) + c; t[i] = (uint32_t)v; c = (uint32_t)(v >> 32); }
    for (int i = 0; i < 5; i++) m[i] = t[i];
    return neg;
}
ZKV_HD void glv_split(const uint32_t (&k)[8], uint32_t (&m1)[5], uint32_t& neg1, uint32_t (&m2)[5], uint32_t& neg2) {
    const uint32_t G1[5] = ZKV_GLV_G1, G2[3] = ZKV_GLV_G2, A1[4] = ZKV_GLV_A1, NN[2] = ZKV_GLV_N, B2[4] = ZKV_GLV_B2;
    uint32_t p1[13], p2[11];
    glv_mul<8, 5>(k, G1, p1); glv_mul<8, 3>(k, G2, p2);
    uint32_t c1[5], c2[3];
    for (int i = 0; i < 5; i++) c1[i] = p1[8 + i];
    for (int i = 0; i < 3; i++) c2[i] = p2[8 + i];
    uint32_t t1[9], t2[5], u1[7], u2[7];
    glv_mul<5, 4>(c1, A1, t1); glv_mul<3, 2>(c2, NN, t2); glv_mul<5, 2>(c1, NN, u1); glv_mul<3, 4>(c2, B2, u2);
    neg1 = glv_diff(k, t1, t2, 5, m1);                      // k - c1 a1 - c2 n   (low six words are enough)
    neg2 = glv_diff(u1, u2, nullptr, 0, m2);               // c1 n - c2 b2
}
// k P for ANY 256-bit k (the ecMul precompile, EIP-196: the group has order r, so k is reduced first): the two GLV halves walked jointly, one
// bit of each per step, from the table {P1, P2, P1 + P2} with P1 = +-P, P2 = +-phi(P) carrying the halves' signs -- 131 doublings and
// mixed additions instead of 256 (in a wavefront the addition of a one-bit-per-step loop is executed at every step anyway, so halving
// the steps halves the work), plus one inversion for the affine P1 + P2.  P is affine and not infinity.
ZKV_HD G1J g1_mul_glv(const Fp& x, const Fp& y, const uint32_t kraw[8]) {
    const uint32_t M[8] = ZKV_FR_R_LIMBS;
    uint32_t k[8];
    for (int i = 0; i < 8; i++) k[i] = kraw[i];
#pragma unroll 1
    for (int t = 0; t < 5; t++) {                           // 2^256 < 6 r
        uint32_t d[8], br = 0;
        for (int i = 0; i < 8; i++) d[i] = subb(k[i], M[i], br);
        if (!br) for (int i = 0; i < 8; i++) k[i] = d[i];
    }
    uint32_t m1[5], m2[5], n1, n2;
    glv_split(k, m1, n1, m2, n2);
    const Fp beta = ZKV_GLV_BETA;
    const Fp y1 = n1 ? fp_neg(y) : y, x2 = fp_mul(x, beta), y2 = n2 ? fp_neg(y) : y;
    G1J s; s.x = x; s.y = y1; s.z = fp_one();
    s = g1j_add_affine(s, x2, y2);                          // phi(P) != +-P for P != O: a chord, never infinity
    G1A p3; uint32_t inf3;
    g1j_to_affine(s, p3, inf3);
    G1J acc = g1j_infinity();
#pragma unroll 1
    for (int b = 130; b >= 0; b--) {
        acc = g1j_dbl(acc);
        const uint32_t d = ((m1[b >> 5] >> (b & 31)) & 1u) | (((m2[b >> 5] >> (b & 31)) & 1u) << 1);
        if (d) acc = g1j_add_affine(acc, d == 1 ? x : d == 2 ? x2 : p3.x, d == 1 ? y1 : d == 2 ? y2 : p3.y);
    }
    return acc;
}
}  // namespace zkv
