// Fp12-level operations on memory-resident operands.
//
// Why this layer exists: one proof per lane keeps ~1.5 KB of live tower state per lane.  Letting the
// compiler hold all of it in VGPRs across a fully inlined Miller loop / final exponentiation produces
// multi-hour compiles and thousands of scratch spills on gfx950.  Instead every Fp12 value lives in a
// word-strided memory slot -- LDS for the hot accumulator f and the running point T (lane-interleaved,
// conflict-free: word k of lane l at base[k*64 + l]), HBM struct-of-arrays slots for the few cold values of
// the final exponentiation -- and each Fp12-level operation is one non-inlined function whose temporaries
// (a few Fp6) are register resident.  fp_mul is the non-inlined leaf.
#pragma once
#include "zkv_curve.h"

namespace zkv {

// MRef: generic slot, word k of this lane's value at p[k * stride] (HBM struct-of-arrays, constants, host tests).
// f2w = words between consecutive Fp2 coefficients of the slot: 16 for the full layout (c0 then c1; a paired lane points p
// at its own component), 8 for a lane-private half slot.
struct MRef {
    uint32_t* p; uint32_t stride; uint32_t f2w;
    ZKV_HD uint32_t ld(int k) const { return p[(size_t)k * stride]; }
    ZKV_HD void st(int k, uint32_t v) const { p[(size_t)k * stride] = v; }
    ZKV_HD int fw() const { return (int)f2w; }
};
ZKV_HD MRef m_ref(uint32_t* p, uint32_t stride, uint32_t f2w = 16) { MRef r; r.p = p; r.stride = stride; r.f2w = f2w; return r; }
ZKV_HD MRef m_off(MRef m, int words) { MRef r = m; r.p = m.p + (size_t)words * m.stride; return r; }

// LRef: lane-interleaved LDS slot of a one-wavefront workgroup, word k at p[k * 64].  The pointer is typed as LDS
// (address space 3) and the stride is a compile-time constant, so every access is a ds_read_b32 / ds_write_b32 with an
// immediate offset: no address arithmetic, no flat-address lookup, conflict-free banks.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) uint32_t zkv_lds_u32;
#else
typedef uint32_t zkv_lds_u32;
#endif
struct LRef {
    zkv_lds_u32* p;
    ZKV_HD uint32_t ld(int k) const { return p[k * 64]; }
    ZKV_HD void st(int k, uint32_t v) const { p[k * 64] = v; }
#if defined(ZKV_PAIRED)
    ZKV_HD int fw() const { return 8; }
#else
    ZKV_HD int fw() const { return 16; }
#endif
};
ZKV_HD LRef l_ref(uint32_t* lds_base_plus_lane) { LRef r; r.p = (zkv_lds_u32*)lds_base_plus_lane; return r; }

template <class R> ZKV_HD Fp m_ld_fp(R m, int word0) {
    Fp r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = m.ld(word0 + k);
    return r;
}
template <class R> ZKV_HD void m_st_fp(R m, int word0, const Fp& a) {
#pragma unroll
    for (int k = 0; k < 8; k++) m.st(word0 + k, a.v[k]);
}
#if defined(ZKV_PAIRED)
template <class R> ZKV_HD Fp2 m_ld_f2(R m, int idx) { Fp2 r; r.h = m_ld_fp(m, m.fw() * idx); return r; }
template <class R> ZKV_HD void m_st_f2(R m, int idx, const Fp2& a) { m_st_fp(m, m.fw() * idx, a.h); }
#else
template <class R> ZKV_HD Fp2 m_ld_f2(R m, int idx) { Fp2 r; r.c0 = m_ld_fp(m, 16 * idx); r.c1 = m_ld_fp(m, 16 * idx + 8); return r; }
template <class R> ZKV_HD void m_st_f2(R m, int idx, const Fp2& a) { m_st_fp(m, 16 * idx, a.c0); m_st_fp(m, 16 * idx + 8, a.c1); }
#endif
template <class R> ZKV_HD Fp6 m_ld_f6(R m, int idx) { Fp6 r; r.c0 = m_ld_f2(m, idx); r.c1 = m_ld_f2(m, idx + 1); r.c2 = m_ld_f2(m, idx + 2); return r; }
template <class R> ZKV_HD void m_st_f6(R m, int idx, const Fp6& a) { m_st_f2(m, idx, a.c0); m_st_f2(m, idx + 1, a.c1); m_st_f2(m, idx + 2, a.c2); }
// An Fp12 slot is six Fp2 coefficients g0 g1 g2 h0 h1 h2.
template <class RD, class RA> ZKV_HD void f12m_copy(RD d, RA a) {
#pragma unroll 1
    for (int k = 0; k < 6; k++) m_st_f2(d, k, m_ld_f2(a, k));
}
template <class RD> ZKV_HD void f12m_set_one(RD d) {
    m_st_f2(d, 0, f2_one());
#pragma unroll 1
    for (int k = 1; k < 6; k++) m_st_f2(d, k, f2_zero());
}
template <class RD> ZKV_HD void f12m_conj(RD d) {            // in place: negate h
#pragma unroll 1
    for (int k = 3; k < 6; k++) m_st_f2(d, k, f2_neg(m_ld_f2(d, k)));
}
template <class RA> ZKV_HD bool f12m_is_one(RA a) {
    bool ok = f2_eq(m_ld_f2(a, 0), f2_one());
#pragma unroll 1
    for (int k = 1; k < 6; k++) ok = f2_is_zero(m_ld_f2(a, k)) && ok;
    return ok;
}

// f <- f^2 (complex squaring, 12 Fp2 products)
template <class RF> ZKV_HD void f12m_sqr_body(RF f) {
    Fp6 g = m_ld_f6(f, 0), h = m_ld_f6(f, 3);
    Fp6 t = f6_mul(g, h);
    Fp6 s = f6_mul(f6_add(g, h), f6_add(g, f6_mul_v(h)));
    m_st_f6(f, 0, f6_sub(f6_sub(s, t), f6_mul_v(t)));
    m_st_f6(f, 3, f6_add(t, t));
}
template <class RF> ZKV_HD_NI void f12m_sqr(RF f) { f12m_sqr_body(f); }
// f <- f^2 for f in the cyclotomic subgroup (after the easy part of the final exponentiation):
// Granger-Scott, three Fp4 squarings = 6 Fp2 products instead of 12.
ZKV_HD void fp4_sqr(const Fp2& a, const Fp2& b, Fp2& t0, Fp2& t1) {     // (a + b y)^2, y^2 = xi
    Fp2 tmp = f2_mul(a, b);
    t0 = f2_sub(f2_sub(f2_mul(f2_add(a, b), f2_add(f2_mul_xi(b), a)), tmp), f2_mul_xi(tmp));
    t1 = f2_dbl(tmp);
}
template <class RF> ZKV_HD void f12m_cyclo_sqr_body(RF f) {
    // memory order g0 g1 g2 h0 h1 h2; pairs (g0,h1), (h0,g2), (g1,h2)
    Fp2 z0 = m_ld_f2(f, 0), z1 = m_ld_f2(f, 4), t0, t1;
    fp4_sqr(z0, z1, t0, t1);
    z0 = f2_sub(t0, z0); z0 = f2_add(f2_dbl(z0), t0);           // 3 t0 - 2 z0
    z1 = f2_add(t1, z1); z1 = f2_add(f2_dbl(z1), t1);           // 3 t1 + 2 z1
    m_st_f2(f, 0, z0); m_st_f2(f, 4, z1);
    Fp2 z2 = m_ld_f2(f, 3), z3 = m_ld_f2(f, 2), t2, t3;
    fp4_sqr(z2, z3, t2, t3);
    Fp2 z4 = m_ld_f2(f, 1), z5 = m_ld_f2(f, 5), t4, t5;
    fp4_sqr(z4, z5, t4, t5);
    Fp2 x = f2_mul_xi(t5);
    z2 = f2_add(x, z2); z2 = f2_add(f2_dbl(z2), x);             // 3 xi t5 + 2 z2
    z3 = f2_sub(t4, z3); z3 = f2_add(f2_dbl(z3), t4);           // 3 t4 - 2 z3
    z4 = f2_sub(t2, z4); z4 = f2_add(f2_dbl(z4), t2);           // 3 t2 - 2 z4
    z5 = f2_add(t3, z5); z5 = f2_add(f2_dbl(z5), t3);           // 3 t3 + 2 z5
    m_st_f2(f, 3, z2); m_st_f2(f, 2, z3); m_st_f2(f, 1, z4); m_st_f2(f, 5, z5);
}
template <class RF> ZKV_HD_NI void f12m_cyclo_sqr(RF f) { f12m_cyclo_sqr_body(f); }
#if defined(ZKV_PAIRED)
// ---------------------------------------------------------------- the accumulator in resident 29-bit limbs (zkv_field.h, L9)
// L9Ref: lane-interleaved LDS slot of six Fp2 coefficients, nine words each: limb k of coefficient idx at p[(9 idx + k) * 64].
// Through m_ld_fp / m_st_fp (which pack / unpack) the slot also serves the generic Fp12 routines above and below.
struct L9Ref {
    zkv_lds_u32* p;
    ZKV_HD int fw() const { return 8; }
};
ZKV_HD L9Ref l9_ref(uint32_t* lds_base_plus_lane) { L9Ref r; r.p = (zkv_lds_u32*)lds_base_plus_lane; return r; }
ZKV_HD L9 l9_ld(L9Ref m, int idx) {
    L9 r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.l[k] = m.p[(9 * idx + k) * 64];
    return r;
}
ZKV_HD void l9_st(L9Ref m, int idx, const L9& a) {
#pragma unroll
    for (int k = 0; k < 9; k++) m.p[(9 * idx + k) * 64] = a.l[k];
}
ZKV_HD Fp m_ld_fp(L9Ref m, int word0) { return l9_to_fp(l9_ld(m, word0 >> 3)); }
ZKV_HD void m_st_fp(L9Ref m, int word0, const Fp& a) { l9_st(m, word0 >> 3, l9_from_fp(a)); }

// (t0, t1) = (a + b y)^2 with y^2 = xi (Granger-Scott's Fp4 squaring) and the cyclotomic update in one go:
//   o0 = 3 t0 - 2 za,   o1 = 3 t1 + 2 zb   (xi_t1: o1 = 3 xi t1 + 2 zb),   t0 = a^2 + xi b^2 = (a + b)(a + xi b) - (1 + xi) a b,  t1 = 2 a b.
// Two lane products (tmp = a b, S = (a + b)(a + xi b)) and three one-pass linear combinations: a + xi b for the multiplier of S,
//   o0 = 3 S - 30 tmp -+ 3 tmp' - 2 za  (1 + xi = 10 + u: the even lane needs -(10 tmp.re - tmp.im), the odd lane -(10 tmp.im + tmp.re)),
//   o1 = 6 tmp + 2 zb, or 54 tmp -+ 6 tmp' + 2 zb.
// All inputs normalised and below 2p.
ZKV_HD void l9_fp4_sqr_update(const L9& a, const L9& b, const L9& za, const L9& zb, const bool xi_t1, L9& o0, L9& o1) {
    const bool odd = zkv_parity() != 0;
    L9X ax; L9Y by;
    l9_x(a, ax); l9_y(b, by);
    const L9 tmp = l9_mul(ax, by);
    L9X s1;
#pragma unroll
    for (int i = 0; i < 9; i++) s1.own[i] = a.l[i] + b.l[i];                     // lazy limbs below 2^30: multiplicand only
#pragma unroll
    for (int i = 0; i < 9; i++) s1.par[i] = zkv_partner_u32(s1.own[i]);
    L9 s2;
    {
        // this lane's component of a + xi b = mine(a) + 9 mine(b) -+ other(b): the even lane takes 8p - b1 (by.V), the odd lane b0 (by.U)
        uint32_t ob[9];
#pragma unroll
        for (int i = 0; i < 9; i++) ob[i] = odd ? by.U[i] : by.V[i];
        const LTerm t[3] = {{a.l, 1}, {b.l, 9}, {ob, 1}};
        s2 = l9_lincomb(t, 1);
    }
    L9Y s2y; l9_y(s2, s2y);
    const L9 S = l9_mul(s1, s2y);
    const L9 tp = l9_partner(tmp);
    const int32_t k3 = odd ? -3 : 3;
    {
        const LTerm t[4] = {{S.l, 3}, {tmp.l, -30}, {tp.l, k3}, {za.l, -2}};
        o0 = l9_lincomb(t, 72);                                                    // 1 + (30 + 3 + 2) * 2
    }
    if (!xi_t1) {
        const LTerm t[2] = {{tmp.l, 6}, {zb.l, 2}};
        o1 = l9_lincomb(t, 1);
    } else {
        const LTerm t[3] = {{tmp.l, 54}, {tp.l, -2 * k3}, {zb.l, 2}};
        o1 = l9_lincomb(t, 14);                                                    // 1 + 6 * 2
    }
}
// f <- f^2 for f in the cyclotomic subgroup, accumulator in resident limbs.  Memory order g0 g1 g2 h0 h1 h2; pairs (g0,h1), (h0,g2), (g1,h2):
//   g0' = 3 t0 - 2 g0, h1' = 3 t1 + 2 h1;   g1' = 3 t2 - 2 g1, h2' = 3 t3 + 2 h2 with (t2, t3) from (h0, g2);
//   g2' = 3 t4 - 2 g2, h0' = 3 xi t5 + 2 h0 with (t4, t5) from (g1, h2).
ZKV_HD void f12l9_cyclo_sqr(L9Ref f) {
    {
        const L9 a = l9_ld(f, 0), b = l9_ld(f, 4);
        L9 o0, o1;
        l9_fp4_sqr_update(a, b, a, b, false, o0, o1);
        l9_st(f, 0, o0); l9_st(f, 4, o1);
    }
    L9 n1, n5;
    {
        const L9 a = l9_ld(f, 3), b = l9_ld(f, 2), za = l9_ld(f, 1), zb = l9_ld(f, 5);
        l9_fp4_sqr_update(a, b, za, zb, false, n1, n5);
    }
    {
        const L9 a = l9_ld(f, 1), b = l9_ld(f, 5), za = l9_ld(f, 2), zb = l9_ld(f, 3);
        L9 o0, o1;
        l9_fp4_sqr_update(a, b, za, zb, true, o0, o1);
        l9_st(f, 2, o0); l9_st(f, 3, o1);
    }
    l9_st(f, 1, n1); l9_st(f, 5, n5);
}
#endif  // ZKV_PAIRED

// d <- a * b, or a * conj(b) (conj(b) = b^-1 for b in the cyclotomic subgroup); d may alias a or b
template <class RD, class RA, class RB> ZKV_HD void f12m_mul_body(RD d, RA a, RB b, bool conj_b) {
    Fp6 ag = m_ld_f6(a, 0), bg = m_ld_f6(b, 0);
    Fp6 t0 = f6_mul(ag, bg);
    Fp6 ah = m_ld_f6(a, 3), bh = m_ld_f6(b, 3);
    if (conj_b) bh = f6_neg(bh);
    Fp6 t1 = f6_mul(ah, bh);
    Fp6 m = f6_mul(f6_add(ag, ah), f6_add(bg, bh));
    m_st_f6(d, 3, f6_sub(f6_sub(m, t0), t1));
    m_st_f6(d, 0, f6_add(t0, f6_mul_v(t1)));
}
template <class RD, class RA, class RB> ZKV_HD_NI void f12m_mul(RD d, RA a, RB b) { f12m_mul_body(d, a, b, false); }
template <class RD, class RA, class RB> ZKV_HD_NI void f12m_mul_conj(RD d, RA a, RB b) { f12m_mul_body(d, a, b, true); }
// f <- f * (c0 + (c3 + c4 v) w)
#if defined(ZKV_PAIRED)
// Lane pairs: the thirteen Karatsuba products inlined, with the multipliers c0 and c4 (three and four products) and the multiplicands
// h2 and g2 + h2 (two each) unpacked and exchanged once: k_miller2 112.5 -> 110.7 ms.  (Keeping c3 and c0 + c3 prepared as well needs
// 36 more registers and spilled: 115.9 ms.)
template <class RF> ZKV_HD void f12m_mul_by_034_body(RF f, const Fp2& c0, const Fp2& c3, const Fp2& c4) {
    uint32_t c0U[9], c0V[9], c4U[9], c4V[9], tU[9], tV[9], xo[9], xp[9];
    f2_limbs_y(c0.h, c0U, c0V); f2_limbs_y(c4.h, c4U, c4V);
    Fp2 t0c0, t0c1, t0c2, t1c0, t1c1, t1c2;
    {
        const Fp2 h0 = m_ld_f2(f, 3), h1 = m_ld_f2(f, 4), h2 = m_ld_f2(f, 5);
        Fp2 p4, p5, p6, p7, p8;
        f2_limbs_y(c3.h, tU, tV);
        f2_limbs_x(h0.h, xo, xp); p4.h = f2_mul_limbs(xo, xp, tU, tV);
        f2_limbs_x(h2.h, xo, xp); p8.h = f2_mul_limbs(xo, xp, tU, tV); p7.h = f2_mul_limbs(xo, xp, c4U, c4V);
        f2_limbs_x(h1.h, xo, xp); p5.h = f2_mul_limbs(xo, xp, c4U, c4V);
        f2_limbs_y(f2_add_nr(c3, c4).h, tU, tV);
        f2_limbs_x(f2_add_nr(h0, h1).h, xo, xp); p6.h = f2_mul_limbs(xo, xp, tU, tV);
        t1c0 = f2_add(p4, f2_mul_xi(p7)); t1c1 = f2_sub(f2_sub(p6, p4), p5); t1c2 = f2_add(p5, p8);
    }
    Fp2 s0, s1, s2;
    {
        const Fp2 g0 = m_ld_f2(f, 0), g1 = m_ld_f2(f, 1), g2 = m_ld_f2(f, 2);
        f2_limbs_x(g0.h, xo, xp); t0c0.h = f2_mul_limbs(xo, xp, c0U, c0V);
        f2_limbs_x(g1.h, xo, xp); t0c1.h = f2_mul_limbs(xo, xp, c0U, c0V);
        f2_limbs_x(g2.h, xo, xp); t0c2.h = f2_mul_limbs(xo, xp, c0U, c0V);
        s0 = f2_add(g0, m_ld_f2(f, 3)); s1 = f2_add(g1, m_ld_f2(f, 4)); s2 = f2_add(g2, m_ld_f2(f, 5));
    }
    m_st_f2(f, 0, f2_add(t0c0, f2_mul_xi(t1c2))); m_st_f2(f, 1, f2_add(t0c1, t1c0)); m_st_f2(f, 2, f2_add(t0c2, t1c1));
    const Fp2 b0 = f2_add(c0, c3);
    Fp2 w0, w1, n, q;
    f2_limbs_y(b0.h, tU, tV);
    f2_limbs_x(s0.h, xo, xp); w0.h = f2_mul_limbs(xo, xp, tU, tV);
    f2_limbs_x(s2.h, xo, xp); q.h = f2_mul_limbs(xo, xp, tU, tV);
    Fp2 q2; q2.h = f2_mul_limbs(xo, xp, c4U, c4V);
    f2_limbs_x(s1.h, xo, xp); w1.h = f2_mul_limbs(xo, xp, c4U, c4V);
    f2_limbs_y(f2_add_nr(b0, c4).h, tU, tV);
    f2_limbs_x(f2_add_nr(s0, s1).h, xo, xp); n.h = f2_mul_limbs(xo, xp, tU, tV);
    m_st_f2(f, 3, f2_sub(f2_sub(f2_add(w0, f2_mul_xi(q2)), t0c0), t1c0));
    m_st_f2(f, 4, f2_sub(f2_sub(f2_sub(f2_sub(n, w0), w1), t0c1), t1c1));
    m_st_f2(f, 5, f2_sub(f2_sub(f2_add(w1, q), t0c2), t1c2));
}
#else
template <class RF> ZKV_HD void f12m_mul_by_034_body(RF f, const Fp2& c0, const Fp2& c3, const Fp2& c4) {
    Fp6 g = m_ld_f6(f, 0), h = m_ld_f6(f, 3);
    Fp6 t0 = f6_mul_fp2(g, c0);
    Fp6 t1 = f6_mul_by_01(h, c3, c4);
    Fp6 t2 = f6_mul_by_01(f6_add(g, h), f2_add(c0, c3), c4);
    m_st_f6(f, 3, f6_sub(f6_sub(t2, t0), t1));
    m_st_f6(f, 0, f6_add(t0, f6_mul_v(t1)));
}
#endif
template <class RF> ZKV_HD_NI void f12m_mul_by_034(RF f, const Fp2* c0, const Fp2* c3, const Fp2* c4) { f12m_mul_by_034_body(f, *c0, *c3, *c4); }
// f <- f * (1 + (c3 + c4 v) w)
#if defined(ZKV_PAIRED)
// With l = c3 + c4 v:  g' = g + v (h l),  h' = h + g l.  Written out per coefficient every output is its input plus a sum of TWO
// products -- g'0 = g0 + h1 (xi c4) + h2 (xi c3), g'1 = g1 + h0 c3 + h2 (xi c4), g'2 = g2 + h0 c4 + h1 c3, and the same for h' with
// g in place of h and no wrap for h'0: h'0 = h0 + g0 c3 + g2 (xi c4) -- so the six outputs are six fused two-product sums (one
// reduction each, f2_dot2_limbs) instead of ten products with Karatsuba's additions: the same 2,430 multiplies per lane, but 6 instead
// of 10 reduce / pack sets and 8 instead of 24 modular additions and xi-multiplications (k_miller2 124.2 -> 117.7 ms as six calls of a
// leaf that read its two coefficients of f from LDS).  And every operand is unpacked and exchanged ONCE: the four line coefficients
// (c3, xi c3, c4, xi c4) enter five or six of the twelve products each and every coefficient of f two, so the per-product form spent
// 500 of its 4,700 instructions unpacking and exchanging values it had already seen; inlined into the caller's loop body
// (117.7 -> 112.5 ms).
template <class RF> ZKV_HD void f12m_mul_by_134_body(RF f, const Fp2& c3, const Fp2& c4) {
    uint32_t c3U[9], c3V[9], c4U[9], c4V[9], x3U[9], x3V[9], x4U[9], x4V[9];
    f2_limbs_y(c3.h, c3U, c3V); f2_limbs_y(c4.h, c4U, c4V);
    { const Fp2 x3 = f2_mul_xi(c3), x4 = f2_mul_xi(c4); f2_limbs_y(x3.h, x3U, x3V); f2_limbs_y(x4.h, x4U, x4V); }
    Fp ng0, ng1, ng2, nh0, nh1, nh2;
    {
        uint32_t h0o[9], h0p[9], h1o[9], h1p[9], h2o[9], h2p[9];
        f2_limbs_x(m_ld_f2(f, 3).h, h0o, h0p); f2_limbs_x(m_ld_f2(f, 4).h, h1o, h1p); f2_limbs_x(m_ld_f2(f, 5).h, h2o, h2p);
        ng0 = fp_add(m_ld_f2(f, 0).h, f2_dot2_limbs(h1o, h1p, x4U, x4V, h2o, h2p, x3U, x3V));
        ng1 = fp_add(m_ld_f2(f, 1).h, f2_dot2_limbs(h0o, h0p, c3U, c3V, h2o, h2p, x4U, x4V));
        ng2 = fp_add(m_ld_f2(f, 2).h, f2_dot2_limbs(h0o, h0p, c4U, c4V, h1o, h1p, c3U, c3V));
    }
    {
        uint32_t g0o[9], g0p[9], g1o[9], g1p[9], g2o[9], g2p[9];
        f2_limbs_x(m_ld_f2(f, 0).h, g0o, g0p); f2_limbs_x(m_ld_f2(f, 1).h, g1o, g1p); f2_limbs_x(m_ld_f2(f, 2).h, g2o, g2p);
        nh0 = fp_add(m_ld_f2(f, 3).h, f2_dot2_limbs(g0o, g0p, c3U, c3V, g2o, g2p, x4U, x4V));
        nh1 = fp_add(m_ld_f2(f, 4).h, f2_dot2_limbs(g0o, g0p, c4U, c4V, g1o, g1p, c3U, c3V));
        nh2 = fp_add(m_ld_f2(f, 5).h, f2_dot2_limbs(g1o, g1p, c4U, c4V, g2o, g2p, c3U, c3V));
    }
    Fp2 o;
    o.h = ng0; m_st_f2(f, 0, o); o.h = ng1; m_st_f2(f, 1, o); o.h = ng2; m_st_f2(f, 2, o);
    o.h = nh0; m_st_f2(f, 3, o); o.h = nh1; m_st_f2(f, 4, o); o.h = nh2; m_st_f2(f, 5, o);
}
#else
// one proof per lane (set-up kernels, host reference, op count of the canonical algorithm): two Karatsuba products by c3 + c4 v
template <class RF> ZKV_HD void f12m_mul_by_134_body(RF f, const Fp2& c3, const Fp2& c4) {
    Fp6 g = m_ld_f6(f, 0), h = m_ld_f6(f, 3);
    Fp6 hs = f6_mul_by_01(h, c3, c4);
    Fp6 gs = f6_mul_by_01(g, c3, c4);
    m_st_f6(f, 0, f6_add(g, f6_mul_v(hs)));
    m_st_f6(f, 3, f6_add(h, gs));
}
#endif
template <class RF> ZKV_HD_NI void f12m_mul_by_134(RF f, const Fp2* c3, const Fp2* c4) { f12m_mul_by_134_body(f, *c3, *c4); }
// d <- a^-1
template <class RD, class RA> ZKV_HD void f12m_inv_body(RD d, RA a) {
    Fp6 g = m_ld_f6(a, 0), h = m_ld_f6(a, 3);
    Fp6 t = f6_sub(f6_mul(g, g), f6_mul_v(f6_mul(h, h)));
    t = f6_inv(t);
    m_st_f6(d, 0, f6_mul(g, t));
    m_st_f6(d, 3, f6_neg(f6_mul(h, t)));
}
template <class RD, class RA> ZKV_HD_NI void f12m_inv(RD d, RA a) { f12m_inv_body(d, a); }
// d <- pi^k(a), k = 1, 2, 3 (d may alias a)
ZKV_TABLE Fp2C ZKV_FROB1_TAB[6] = ZKV_FROB1;
ZKV_TABLE Fp ZKV_FROB2_TAB[6] = ZKV_FROB2;
ZKV_TABLE Fp2C ZKV_FROB3_TAB[6] = ZKV_FROB3;
template <class RD, class RA> ZKV_HD void f12m_frob_body(RD d, RA a, int k) {
    const Fp2C* G1 = ZKV_FROB1_TAB; const Fp* G2 = ZKV_FROB2_TAB; const Fp2C* G3 = ZKV_FROB3_TAB;
    // memory order g0 g1 g2 h0 h1 h2 <-> w-powers 0 2 4 1 3 5
    const int wp[6] = {0, 2, 4, 1, 3, 5};
#pragma unroll 1
    for (int i = 0; i < 6; i++) {
        Fp2 c = m_ld_f2(a, i);
        int e = wp[i];
        if (k == 2) c = f2_mul_fp(c, G2[e]);
        else {
            c = f2_conj(c);
            if (e) c = f2_mul(c, f2_const(k == 1 ? G1[e] : G3[e]));
        }
        m_st_f2(d, i, c);
    }
}
template <class RD, class RA> ZKV_HD_NI void f12m_frob(RD d, RA a, int k) { f12m_frob_body(d, a, k); }

// T <- 2T with tangent-line coefficients (T is 3 Fp2 in memory)
template <class RT> ZKV_HD_NI void g2m_line_dbl(RT Tm, Fp2* l0, Fp2* l1, Fp2* l3) {
    G2H T; T.x = m_ld_f2(Tm, 0); T.y = m_ld_f2(Tm, 1); T.z = m_ld_f2(Tm, 2);
    line_dbl(T, *l0, *l1, *l3);
    m_st_f2(Tm, 0, T.x); m_st_f2(Tm, 1, T.y); m_st_f2(Tm, 2, T.z);
}
template <class RT> ZKV_HD_NI void g2m_line_add(RT Tm, const Fp2* qx, const Fp2* qy, Fp2* l0, Fp2* l1, Fp2* l3) {
    G2H T; T.x = m_ld_f2(Tm, 0); T.y = m_ld_f2(Tm, 1); T.z = m_ld_f2(Tm, 2);
    line_add(T, *qx, *qy, *l0, *l1, *l3);
    m_st_f2(Tm, 0, T.x); m_st_f2(Tm, 1, T.y); m_st_f2(Tm, 2, T.z);
}

}  // namespace zkv
