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

// Writable rows addressed like SoaRef (wave-uniform base and stride, one 32-bit byte offset per lane); usable wherever an MRef is:
// Fp2 value k of this lane at words 16 k .. 16 k + 7 (the lane offset already selects the lane's component).
struct SoaRW {
    uint32_t* p; size_t stride; uint32_t off;
    ZKV_HD uint32_t ld(int k) const { return *(const uint32_t*)((const char*)(p + (size_t)k * stride) + off); }
    ZKV_HD void st(int k, uint32_t v) const { *(uint32_t*)((char*)(p + (size_t)k * stride) + off) = v; }
    ZKV_HD int fw() const { return 16; }
};
// m_fresh(ref): the same reference with its per-lane part made opaque to the compiler (SoaRW only; a no-op for the other kinds).  Placed
// in front of a group of loads or stores it keeps the compiler from computing their addresses far ahead and parking them in scratch.
template <class R> ZKV_HD R m_fresh(R r) { return r; }
ZKV_HD SoaRW m_fresh(SoaRW r) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(r.off));
#endif
    return r;
}

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
    const int32_t k1 = odd ? 1 : -1, k3 = odd ? -3 : 3;
    const L9 tmp = l9_mul(a, b);
    L9 s1, s2;
#pragma unroll
    for (int i = 0; i < 9; i++) s1.l[i] = a.l[i] + b.l[i];                       // lazy limbs below 2^30: multiplicand only
    {
        // this lane's component of a + xi b = mine(a) + 9 mine(b) -+ other(b)
        const L9 pb = l9_partner(b);
        const LTerm t[3] = {{a.l, 1}, {b.l, 9}, {pb.l, k1}};
        s2 = l9_lincomb(t, 4);
    }
    const L9 S = l9_mul(s1, s2);
    const L9 tp = l9_partner(tmp);
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
// ---- ACC <- ACC * S (or ACC * conj(S)) on resident limbs, S packed in an HBM slot
// S is a row reference with a 32-bit lane offset (SoaRW); each group of loads starts from m_fresh(S).
ZKV_HD void l9_carry(uint32_t (&x)[9]) {            // exact normalisation of lazy non-negative limbs (the value does not change)
#pragma unroll
    for (int k = 0; k < 8; k++) { x[k + 1] += x[k] >> 29; x[k] &= 0x1fffffffu; }
}
// The six Karatsuba products of an Fp6 multiplication (a0 + a1 v + a2 v^2)(b0 + b1 v + b2 v^2) and the limb-wise (lazy, signed) sums
// its three result coefficients are made of:   c0 = v0 + xi wa,   c1 = wb + xi v2,   c2 = wc   with
//   wa = (a1 + a2)(b1 + b2) - v1 - v2,   wb = (a0 + a1)(b0 + b1) - v0 - v1,   wc = (a0 + a2)(b0 + b2) - v0 - v2 + v1.
// The caller turns them into reduced coefficients (l9_lincomb), together with whatever else it adds to them.
// lda(i) / ldb(i): this lane's limbs of coefficient i of the multiplicand / multiplier (normalised, values below 2p: the sum of two
// multiplicands stays lazy, the sum of two multipliers is carried to normalised limbs).
struct L9F6Raw { L9 v0, v2; int32_t wa[9], wb[9], wc[9]; };
template <class LDA, class LDB> ZKV_HD void l9_f6_mul_raw(LDA lda, LDB ldb, L9F6Raw& r) {
    const L9 b0 = ldb(0), b1 = ldb(1), b2 = ldb(2);
    r.v0 = l9_mul(lda(0), b0);
    const L9 v1 = l9_mul(lda(1), b1);
    r.v2 = l9_mul(lda(2), b2);
    L9 sa, sb, m;
    // the multiplicands are read again where they are needed (LDS) instead of living in registers across the products
    { const L9 a1 = lda(1), a2 = lda(2);
#pragma unroll
      for (int i = 0; i < 9; i++) { sa.l[i] = a1.l[i] + a2.l[i]; sb.l[i] = b1.l[i] + b2.l[i]; } }
    l9_carry(sb.l); m = l9_mul(sa, sb);
#pragma unroll
    for (int i = 0; i < 9; i++) r.wa[i] = (int32_t)m.l[i] - (int32_t)v1.l[i] - (int32_t)r.v2.l[i];
    { const L9 a0 = lda(0), a1 = lda(1);
#pragma unroll
      for (int i = 0; i < 9; i++) { sa.l[i] = a0.l[i] + a1.l[i]; sb.l[i] = b0.l[i] + b1.l[i]; } }
    l9_carry(sb.l); m = l9_mul(sa, sb);
#pragma unroll
    for (int i = 0; i < 9; i++) r.wb[i] = (int32_t)m.l[i] - (int32_t)r.v0.l[i] - (int32_t)v1.l[i];
    { const L9 a0 = lda(0), a2 = lda(2);
#pragma unroll
      for (int i = 0; i < 9; i++) { sa.l[i] = a0.l[i] + a2.l[i]; sb.l[i] = b0.l[i] + b2.l[i]; } }
    l9_carry(sb.l); m = l9_mul(sa, sb);
#pragma unroll
    for (int i = 0; i < 9; i++) r.wc[i] = (int32_t)m.l[i] - (int32_t)r.v0.l[i] - (int32_t)r.v2.l[i] + (int32_t)v1.l[i];
}
ZKV_HD void l9_partner_i32(const int32_t (&a)[9], int32_t (&o)[9]) {
#pragma unroll
    for (int i = 0; i < 9; i++) o[i] = (int32_t)zkv_partner_u32((uint32_t)a[i]);
}
// Karatsuba over Fp6: X = ag bg, Y = ah bh, Z = (ag + ah)(bg + bh);  g' = X + v Y,  h' = Z - X - Y.  X and Y are reduced coefficient
// by coefficient (three one-pass combinations each), g' comes from those, and h' is formed in the same passes that reduce Z.
template <class RB> ZKV_HD void f12l9_mul(L9Ref acc, RB S0, const bool conj_b) {
    const int32_t k1 = zkv_parity() != 0 ? 1 : -1;                                 // xi q, this lane's component: 9 mine -+ the partner's
    L9F6Raw raw;
    int32_t pw[9];
    L9 xc0, xc1, xc2;
    {
        const RB S = m_fresh(S0);
        l9_f6_mul_raw([&](int i) { return l9_ld(acc, i); }, [&](int i) { return l9_from_fp(m_ld_f2(S, i).h); }, raw);
        l9_partner_i32(raw.wa, pw);
        { const LTerm t[3] = {{raw.v0.l, 1}, {(const uint32_t*)raw.wa, 9}, {(const uint32_t*)pw, k1}}; xc0 = l9_lincomb(t, 48); }
        const L9 pv2 = l9_partner(raw.v2);
        { const LTerm t[3] = {{(const uint32_t*)raw.wb, 1}, {raw.v2.l, 9}, {pv2.l, k1}}; xc1 = l9_lincomb(t, 8); }
        { const LTerm t[1] = {{(const uint32_t*)raw.wc, 1}}; xc2 = l9_lincomb(t, 8); }
    }
    uint32_t xy0[9], xy1[9], xy2[9];
    {
        L9 g0, g1, g2;
        const RB S = m_fresh(S0);
        l9_f6_mul_raw([&](int i) { return l9_ld(acc, 3 + i); }, [&](int i) {
            Fp2 b = m_ld_f2(S, 3 + i);
            if (conj_b) b = f2_neg(b);
            return l9_from_fp(b.h);
        }, raw);
        L9 yc0, yc1, yc2;
        l9_partner_i32(raw.wa, pw);
        { const LTerm t[3] = {{raw.v0.l, 1}, {(const uint32_t*)raw.wa, 9}, {(const uint32_t*)pw, k1}}; yc0 = l9_lincomb(t, 48); }
        const L9 pv2 = l9_partner(raw.v2);
        { const LTerm t[3] = {{(const uint32_t*)raw.wb, 1}, {raw.v2.l, 9}, {pv2.l, k1}}; yc1 = l9_lincomb(t, 8); }
        { const LTerm t[1] = {{(const uint32_t*)raw.wc, 1}}; yc2 = l9_lincomb(t, 8); }
        const L9 pyc2 = l9_partner(yc2);
        { const LTerm t[3] = {{xc0.l, 1}, {yc2.l, 9}, {pyc2.l, k1}}; g0 = l9_lincomb(t, 4); }
        { const LTerm t[2] = {{xc1.l, 1}, {yc0.l, 1}}; g1 = l9_lincomb(t, 1); }
        { const LTerm t[2] = {{xc2.l, 1}, {yc1.l, 1}}; g2 = l9_lincomb(t, 1); }
#pragma unroll
        for (int i = 0; i < 9; i++) { xy0[i] = xc0.l[i] + yc0.l[i]; xy1[i] = xc1.l[i] + yc1.l[i]; xy2[i] = xc2.l[i] + yc2.l[i]; }
        // g' is final; the slots of h take the multiplicands of Z, ag + ah carried to normalised limbs (values below 4p), until h' replaces them
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const L9 g = l9_ld(acc, i), h = l9_ld(acc, 3 + i);
            L9 sum;
#pragma unroll
            for (int k = 0; k < 9; k++) sum.l[k] = g.l[k] + h.l[k];
            l9_carry(sum.l);
            l9_st(acc, 3 + i, sum);
        }
        l9_st(acc, 0, g0); l9_st(acc, 1, g1); l9_st(acc, 2, g2);
    }
    L9 h0, h1, h2;
    {
        const RB S = m_fresh(S0);
        l9_f6_mul_raw([&](int i) { return l9_ld(acc, 3 + i); }, [&](int i) {
            Fp2 b = m_ld_f2(S, 3 + i);
            if (conj_b) b = f2_neg(b);
            return l9_from_fp(f2_add(m_ld_f2(S, i), b).h);
        }, raw);
        l9_partner_i32(raw.wa, pw);
        { const LTerm t[4] = {{raw.v0.l, 1}, {(const uint32_t*)raw.wa, 9}, {(const uint32_t*)pw, k1}, {xy0, -1}}; h0 = l9_lincomb(t, 52); }
        const L9 pv2 = l9_partner(raw.v2);
        { const LTerm t[4] = {{(const uint32_t*)raw.wb, 1}, {raw.v2.l, 9}, {pv2.l, k1}, {xy1, -1}}; h1 = l9_lincomb(t, 12); }
        { const LTerm t[2] = {{(const uint32_t*)raw.wc, 1}, {xy2, -1}}; h2 = l9_lincomb(t, 12); }
    }
    l9_st(acc, 3, h0); l9_st(acc, 4, h1); l9_st(acc, 5, h2);
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
    d = m_fresh(d);
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
    d = m_fresh(d);
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
