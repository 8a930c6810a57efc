// Coefficient-parallel ("wide") Fp12 arithmetic for small batches: ONE PROOF PER 16 LANES.
//
// In the lane-pair kernels a proof owns one lane pair and walks the 18 Fp2 products of an Fp12 multiplication one after
// the other, so its latency (12 ms) does not depend on the batch size and a batch below 2^16 proofs leaves most of the
// chip idle.  Here the six Fp2 coefficients of every Fp12 value (memory order g0 g1 g2 h0 h1 h2 = powers 0 2 4 1 3 5 of
// w, w^6 = xi) are spread over six lane pairs of a 16-lane group: pair q computes coefficient q of every result.  A pair
// is still "even lane = real part, odd lane = imaginary part", so all Fp2 arithmetic of zkv_field.h is reused unchanged;
// only the Fp12-level routines are new.  Operands live in slots every lane of the group can address (LDS, or the HBM
// struct-of-arrays slots of the final exponentiation); because the whole group sits in one wavefront, every load of a
// routine is issued before any of its stores and no barrier is needed.
//   multiplication      c_e = sum_i a_i b_(e-i) xi^[i + (e-i) mod 6 >= 6]   6 Fp2 products per pair instead of 18 in sequence
//   cyclotomic squaring 2 products per pair instead of 6, sparse line products 3 (or 2) instead of 13 (or 10)
// The line functions spread their independent products over the pairs round by round; the one inversion of the final
// exponentiation is executed redundantly by every pair from the same inputs, which needs no data exchange at all.  Pairs 6 and 7 of a group shadow
// pairs 0 and 1 (same loads, same stores).
//
// Round 3: ONE PROOF PER WAVEFRONT (the mapping BASELINE.json's north star names) is the same code with S = 4 SLICES of 16 lanes: slice s
// of pair q forms every S-th term of coefficient q, the partial sums meet in S scratch rows (`red`, LDS) and every slice adds them up --
// an Fp12 multiplication is 2 rounds of Fp2 products instead of 6, squarings and sparse products 1 instead of 4 / 2 / 3, and the
// per-proof scalings of the line coefficients (four Fp products per step for the fixed pairs, two for the variable one) run side by
// side in the PAIRS of a group instead of in sequence on every pair (both mappings).  S = 1 is the 16-lane kernel (no reduction step is
// compiled in).
#pragma once
#include "zkv_verify.h"

#if defined(ZKV_PAIRED)
// The Fp12-level routines of this file are inlined into the two kernels: as separate functions each kept its cross-call values in
// callee-saved registers and saved / restored them around its body (416 / 784-byte frames) -- latency a lone wavefront cannot hide:
// a single verification 3.82 -> 3.42 ms.  The bodies are small (at most six Fp2 products per pair), so code size is no concern.
#define ZKV_W_NI ZKV_HD

namespace zkv {

ZKV_HD int w_pow(int mi) { return mi < 3 ? 2 * mi : 2 * (mi - 3) + 1; }         // memory index -> power of w
ZKV_HD int w_mem(int pw) { return (pw & 1) ? 3 + (pw >> 1) : (pw >> 1); }       // power of w -> memory index
ZKV_HD Fp2 f2_sel(bool c, const Fp2& if_true, const Fp2& if_false) { Fp2 r; r.h = fp_sel(c, if_true.h, if_false.h); return r; }
// Coefficients written by one pair are read by the others: make the stores of a routine visible to the group before the next
// routine loads (work-group scope: the lanes share one wavefront, so this is a wait for outstanding memory operations).
#if !defined(__HIP_DEVICE_COMPILE__)
void zkv_wide_host_barrier();           // host emulation (tests/host_sim): the lanes of a group are threads
#endif
ZKV_HD void wide_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#else
    zkv_wide_host_barrier();
#endif
}
// Between the loads of a routine and its stores ("all loads before any store": pair q reads every coefficient of a slot that
// another pair is about to overwrite).  On the device the lanes of a group share one wavefront and run in lockstep, but nothing
// stops the COMPILER from sinking a load below another lane group's store, so the point is pinned with a scheduling barrier
// (no instruction is emitted, no code motion across it) plus a work-group fence that orders the memory operations themselves;
// the host emulation needs a real rendezvous here.
ZKV_HD void wide_sync() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
#else
    zkv_wide_host_barrier();
#endif
}

// A lane's place in its proof's group: coefficient q (0..5; pairs 6 and 7 shadow 0 and 1) and slice s (0..S-1).
struct WL { int q, s; };
// Sum over the slices of one Fp2 per (slice, pair): every slice parks its part in row s of `red` (S rows of one Fp12 slot each,
// full layout) and reads all S rows back.  Compiled out for S = 1.
template <int S> ZKV_HD Fp2 w_reduce(MRef red, const Fp2& part, WL w) {
    if (S == 1) return part;
    wide_sync();                                          // the previous readers of these rows are done
    m_st_f2(m_off(red, 96 * w.s), w.q, part);
    wide_fence();
    Fp2 r = m_ld_f2(red, w.q);
#pragma unroll
    for (int k = 1; k < S; k++) r = f2_add(r, m_ld_f2(m_off(red, 96 * k), w.q));
    return r;
}
// N <= 4 values of the form v * k (Fp2 times Fp: one Fp product per lane) in ONE round: pair q forms product q mod N (every pair of a
// proof's group needs all N results, and round 2 had every pair form all of them one after the other), parks it in slot q mod N of the
// line functions' scratch `sc` -- free between a line function and the sparse products -- and reads the N slots back.
template <int N> ZKV_HD void w_scale_many(MRef sc, const Fp2 (&v)[N], const Fp (&k)[N], Fp2 (&out)[N], int q) {
    static_assert(N <= 4, "six pairs per group");
    const int j = q % N;
    Fp2 mv = v[0]; Fp mk = k[0];
#pragma unroll
    for (int t = 1; t < N; t++) { mv = f2_sel(j == t, v[t], mv); mk = fp_sel(j == t, k[t], mk); }
    const Fp2 p = f2_mul_fp(mv, mk);
    wide_sync();
    m_st_f2(sc, j, p);                                    // pairs with the same j (and every slice) write the same value
    wide_fence();
#pragma unroll
    for (int t = 0; t < N; t++) out[t] = m_ld_f2(sc, t);
}

ZKV_HD void w12_set_one(MRef d, int q) { wide_sync(); m_st_f2(d, q, f2_sel(q == 0, f2_one(), f2_zero())); wide_fence(); }
ZKV_HD void w12_copy(MRef d, MRef a, int q) { Fp2 c = m_ld_f2(a, q); wide_sync(); m_st_f2(d, q, c); wide_fence(); }
ZKV_HD void w12_conj(MRef d, int q) {                     // in place: negate the h coefficients
    Fp2 c = m_ld_f2(d, q);
    wide_sync();
    m_st_f2(d, q, f2_sel(q >= 3, f2_neg(c), c));
    wide_fence();
}
// d <- a * b, or a * conj(b) (d may alias a or b).  Slice s takes the terms i = s, s + S, ...: 6 rounds for S = 1, 2 for S = 4.
template <int S> ZKV_W_NI void w12_mul(MRef d, MRef a, MRef b, WL w, bool conj_b, MRef red) {
    const int q = w.q, e = w_pow(q);
    Fp2 accn = f2_zero(), accw = f2_zero();
    const Fp2 zero = f2_zero();
#pragma unroll 1
    for (int k = 0; k < (6 + S - 1) / S; k++) {
        const int it = w.s + k * S;
        const bool valid = it < 6;
        const int i = valid ? it : 0;
        int j = e - i;
        if (j < 0) j += 6;
        const bool wrap = i + j >= 6;
        const int mj = w_mem(j);
        Fp2 x = m_ld_f2(a, w_mem(i)), y = m_ld_f2(b, mj);
        if (conj_b) y = f2_sel(mj >= 3, f2_neg(y), y);
        Fp2 p = f2_mul(x, y);
        accw = f2_add(accw, f2_sel(wrap && valid, p, zero));
        accn = f2_add(accn, f2_sel(wrap || !valid, zero, p));
    }
    const Fp2 r = w_reduce<S>(red, f2_add(accn, f2_mul_xi(accw)), w);
    wide_sync();
    m_st_f2(d, q, r);
    wide_fence();
}
// f <- f^2 (generic): c_e = sum over unordered {i, j}, i + j = e mod 6, of a_i a_j (twice when i != j): at most four products per
// pair.  Table entry = i | j << 3 | doubled << 6 | wrapped << 7 | valid << 8 for output power e, term t (index 4 e + t).
// Round 4: on 29-bit limbs like w12_cyclo_sqr.  The four products of a coefficient (one per slice for S = 4) meet in ONE one-pass combination
// whose per-lane coefficients carry the doubling of the mixed terms and the xi of the wrapped ones.  For S = 4 a slice parks limbs 0..7 of its
// product in its row of `red` and limb 8 in the 48 words that follow the four rows (W_RED_WORDS).
constexpr int W_RED_WORDS = 4 * 96 + 48;
template <int S> ZKV_W_NI void w12_sqr(MRef f, WL wl, MRef red) {
    const uint16_t TERMS[24] = {256, 489, 482, 411, 328, 490, 483, 0, 336, 265, 491, 420, 344, 337, 492, 0, 352, 345, 274, 429, 360, 353, 346, 0};
    const int q = wl.q, e = w_pow(q);
    static_assert(S == 1 || S == 4, "four terms per coefficient: one slice each, or all in one");
    const uint32_t h = zkv_parity();
    const int32_t k1 = h ? 1 : -1;
    L9 p[4];
    int32_t co[4], cp[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const uint32_t w = TERMS[4 * e + t];
        const int32_t two = ((w >> 6) & 1u) ? 2 : 1;
        const bool valid = ((w >> 8) & 1u) != 0, wrap = ((w >> 7) & 1u) != 0;
        co[t] = valid ? two * (wrap ? 9 : 1) : 0;
        cp[t] = (valid && wrap) ? two * k1 : 0;
    }
    if (S == 1) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint32_t w = TERMS[4 * e + t];
            p[t] = l9_mul(l9_from_fp(m_ld_f2(f, w_mem((int)(w & 7u))).h), l9_from_fp(m_ld_f2(f, w_mem((int)((w >> 3) & 7u))).h));
        }
    } else {
        const uint32_t w = TERMS[4 * e + wl.s];
        const L9 mine = l9_mul(l9_from_fp(m_ld_f2(f, w_mem((int)(w & 7u))).h), l9_from_fp(m_ld_f2(f, w_mem((int)((w >> 3) & 7u))).h));
        uint32_t* r8 = red.p - 8 * h + 4 * 96;                    // the ninth limbs: word 12 row + 2 q + half
        wide_sync();
        {
            const MRef rs = m_off(red, 96 * wl.s);
#pragma unroll
            for (int i = 0; i < 8; i++) rs.st(16 * q + i, mine.l[i]);
            r8[12 * wl.s + 2 * q + (int)h] = mine.l[8];
        }
        wide_fence();
#pragma unroll
        for (int t = 0; t < 4; t++) {
#pragma unroll
            for (int i = 0; i < 8; i++) p[t].l[i] = m_off(red, 96 * t).ld(16 * q + i);
            p[t].l[8] = r8[12 * t + 2 * q + (int)h];
        }
    }
    const L9 pp0 = l9_partner(p[0]), pp1 = l9_partner(p[1]), pp2 = l9_partner(p[2]), pp3 = l9_partner(p[3]);
    const LTerm tm[8] = {{p[0].l, co[0]}, {p[1].l, co[1]}, {p[2].l, co[2]}, {p[3].l, co[3]}, {pp0.l, cp[0]}, {pp1.l, cp[1]}, {pp2.l, cp[2]}, {pp3.l, cp[3]}};
    Fp2 r; r.h = l9_to_fp(l9_lincomb(tm, 20));
    wide_sync();
    m_st_f2(f, q, r);
    wide_fence();
}
// f <- f^2 for f in the cyclotomic subgroup (Granger-Scott): pair q needs one Fp4 squaring (A + B y)^2, y^2 = xi.
// S = 4: the two products of the Fp4 squaring (A B and (A + B)(xi B + A)) are formed by the even and the odd slices side by side.
// Round 4: the arithmetic of a step runs on 29-bit limbs (zkv_field.h "L9"): the three coefficients are unpacked once, the two operands of
// the second product and the WHOLE result -- 3 (s - (1 + xi) ab) - 2z, or 6 ab + 2z, or 3 xi (2 ab) + 2z for pair 3 -- are one-pass linear
// combinations with per-lane coefficients (the instruction stream stays uniform), and the result is packed once.  A step was two
// xi-multiplications and three modular additions in front of the product and two xi-multiplications and eight additions behind it, each
// a carry chain: about 900 instructions on the 189-step dependent chain of a single verification, now about 650.
template <int S> ZKV_W_NI void w12_cyclo_sqr(MRef f, WL w, MRef red) {
    const int q = w.q;
    const int ia = (q == 0 || q == 4) ? 0 : (q == 2 || q == 3) ? 1 : 3;
    const int ib = ia == 0 ? 4 : ia == 1 ? 5 : 2;
    const bool odd = q >= 3;                              // this pair's result uses 2AB, otherwise A^2 + xi B^2
    const L9 A = l9_from_fp(m_ld_f2(f, ia).h), B = l9_from_fp(m_ld_f2(f, ib).h), z = l9_from_fp(m_ld_f2(f, q).h);
    const bool im = zkv_parity() != 0;
    const int32_t k1 = im ? 1 : -1, k3 = im ? -3 : 3;
    L9 s1, s2;
#pragma unroll
    for (int i = 0; i < 9; i++) s1.l[i] = A.l[i] + B.l[i];               // lazy limbs: multiplicand only
    {
        const L9 pB = l9_partner(B);
        const LTerm t[3] = {{A.l, 1}, {B.l, 9}, {pB.l, k1}};              // this lane's component of A + xi B
        s2 = l9_lincomb(t, 4);
    }
    L9 ab, sp;
    if (S == 1) {
        ab = l9_mul(A, B);
        sp = l9_mul(s1, s2);
    } else {
        const bool second = (w.s & 1) != 0;
        L9 x, y;
#pragma unroll
        for (int i = 0; i < 9; i++) { x.l[i] = second ? s1.l[i] : A.l[i]; y.l[i] = second ? s2.l[i] : B.l[i]; }
        const L9 p = l9_mul(x, y);
        // slices 0 and 2 (1 and 3) write the same value: limbs 0..7 in row 0 (1) of `red`, limb 8 in word 0 of row 2 (3)
        const MRef r0 = m_off(red, 96 * (w.s & 1)), r8 = m_off(red, 96 * (2 + (w.s & 1)));
        wide_sync();
#pragma unroll
        for (int i = 0; i < 8; i++) r0.st(16 * q + i, p.l[i]);
        r8.st(16 * q, p.l[8]);
        wide_fence();
#pragma unroll
        for (int i = 0; i < 8; i++) { ab.l[i] = red.ld(16 * q + i); sp.l[i] = m_off(red, 96).ld(16 * q + i); }
        ab.l[8] = m_off(red, 192).ld(16 * q); sp.l[8] = m_off(red, 288).ld(16 * q);
    }
    // even pairs: 3 (s - 10 ab +- ab') - 2 z;  odd pairs: 6 ab + 2 z;  pair 3: 3 xi (2 ab) + 2 z = 54 ab -+ 6 ab' + 2 z
    const L9 pab = l9_partner(ab);
    const int32_t cS = odd ? 0 : 3, cAB = odd ? (q == 3 ? 54 : 6) : -30, cP = odd ? (q == 3 ? -2 * k3 : 0) : k3, cZ = odd ? 2 : -2;
    const LTerm t[4] = {{sp.l, cS}, {ab.l, cAB}, {pab.l, cP}, {z.l, cZ}};
    const L9 out = l9_lincomb(t, 72);
    Fp2 o; o.h = l9_to_fp(out);
    wide_sync();
    m_st_f2(f, q, o);
    wide_fence();
}
// f <- f * (c0 + c3 w + c4 w^3); with `one` the constant coefficient is 1 and c0 is not read
// S = 4: slice 0 forms a0 c0 (or passes a0 on), slice 1 a1 c3, slice 2 a3 c4, slice 3 contributes zero: one round.
// Round 4: on 29-bit limbs like w12_cyclo_sqr -- the (at most three) products of a coefficient meet in ONE one-pass combination whose
// per-lane coefficients carry the xi of the wrapped terms (9 mine -+ the partner's), instead of an xi-multiplication computed by every
// lane and selected, and three modular additions.
template <int S> ZKV_W_NI void w12_mul_sparse(MRef f, const Fp2* c0, const Fp2* c3, const Fp2* c4, WL w, bool one, MRef red) {
    const int q = w.q, e = w_pow(q);
    const int e1 = e >= 1 ? e - 1 : e + 5, e3 = e >= 3 ? e - 3 : e + 3;
    const Fp2 a0 = m_ld_f2(f, q), a1 = m_ld_f2(f, w_mem(e1)), a3 = m_ld_f2(f, w_mem(e3));
    const int32_t k1 = zkv_parity() != 0 ? 1 : -1;
    const int32_t k3o = e < 1 ? 9 : 1, k3p = e < 1 ? k1 : 0, k4o = e < 3 ? 9 : 1, k4p = e < 3 ? k1 : 0;        // xi on the terms that wrapped past w^6
    L9 t, p3, p4;
    if (S == 1) {
        t = one ? l9_from_fp(a0.h) : l9_mul(l9_from_fp(a0.h), l9_from_fp(c0->h));
        p3 = l9_mul(l9_from_fp(a1.h), l9_from_fp(c3->h));
        p4 = l9_mul(l9_from_fp(a3.h), l9_from_fp(c4->h));
    } else {
        const Fp2 x = f2_sel(w.s == 1, a1, f2_sel(w.s == 2, a3, a0));
        const Fp2 y = f2_sel(w.s == 1, *c3, f2_sel(w.s == 2, *c4, *c0));          // with `one`, c0 aliases c3: slice 0's product is dropped
        const L9 xl = l9_from_fp(x.h);
        L9 p = l9_mul(xl, l9_from_fp(y.h));
#pragma unroll
        for (int i = 0; i < 9; i++) p.l[i] = (w.s == 0 && one) ? xl.l[i] : p.l[i];
        // slices 0..2 park their product: limbs 0..7 in row s of `red`, limb 8 in word s of row 3 (slice 3 has no term)
        wide_sync();
        if (w.s < 3) {
            const MRef rs = m_off(red, 96 * w.s);
#pragma unroll
            for (int i = 0; i < 8; i++) rs.st(16 * q + i, p.l[i]);
            m_off(red, 288).st(16 * q + w.s, p.l[8]);
        }
        wide_fence();
#pragma unroll
        for (int i = 0; i < 8; i++) { t.l[i] = red.ld(16 * q + i); p3.l[i] = m_off(red, 96).ld(16 * q + i); p4.l[i] = m_off(red, 192).ld(16 * q + i); }
        t.l[8] = m_off(red, 288).ld(16 * q); p3.l[8] = m_off(red, 288).ld(16 * q + 1); p4.l[8] = m_off(red, 288).ld(16 * q + 2);
    }
    const L9 pp3 = l9_partner(p3), pp4 = l9_partner(p4);
    const LTerm tm[5] = {{t.l, 1}, {p3.l, k3o}, {pp3.l, k3p}, {p4.l, k4o}, {pp4.l, k4p}};
    Fp2 r; r.h = l9_to_fp(l9_lincomb(tm, 8));
    wide_sync();
    m_st_f2(f, q, r);
    wide_fence();
}
// d <- pi^k(a), k = 1, 2, 3
ZKV_W_NI void w12_frob(MRef d, MRef a, int k, int q) {
    const Fp2C G1[6] = ZKV_FROB1;
    const Fp G2[6] = ZKV_FROB2;
    const Fp2C G3[6] = ZKV_FROB3;
    const int e = w_pow(q);
    Fp2 c = m_ld_f2(a, q);
    if (k == 2) c = f2_mul_fp(c, G2[e]);
    else c = f2_mul(f2_conj(c), f2_const(k == 1 ? G1[e] : G3[e]));      // the constant of power 0 is 1
    wide_sync();
    m_st_f2(d, q, c);
    wide_fence();
}

// ---------------------------------------------------------------- line functions, products spread over the pairs
// Same formulas as line_dbl / line_add (zkv_curve.h).  Each round every pair forms ONE of the independent Fp2 products of
// that round (its two operands picked with selects, so the instruction stream stays uniform), parks it in the group's
// scratch slots `sc` (full layout, 13 Fp2) and after the fence every pair reads what it needs; the few additions in between
// are cheap and done redundantly.  3 rounds instead of 10 products in sequence for the tangent, 4 instead of 13 for the chord.
ZKV_HD Fp2 w_pick4(int r, const Fp2& a0, const Fp2& a1, const Fp2& a2, const Fp2& a3) {
    return f2_sel(r == 0, a0, f2_sel(r == 1, a1, f2_sel(r == 2, a2, a3)));
}
ZKV_W_NI void w_line_dbl(MRef Tm, MRef sc, Fp2* l0, Fp2* l1, Fp2* l3, int q) {
    const Fp2C b3c = ZKV_TWIST_3B;
    const Fp2 b3 = f2_const(b3c);
    const Fp2 x = m_ld_f2(Tm, 0), y = m_ld_f2(Tm, 1), z = m_ld_f2(Tm, 2);
    {   // round 1: x y | y^2 | z^2 | (y+z)^2 | x^2 (pair 5 repeats pair 4)
        const Fp2 yz = f2_add(y, z);
        const int r = q > 4 ? 4 : q;
        const Fp2 A = f2_sel(r == 4, x, w_pick4(r, x, y, z, yz));
        const Fp2 B = f2_sel(r == 4, x, w_pick4(r, y, y, z, yz));
        const Fp2 p = f2_mul(A, B);
        wide_sync();
        m_st_f2(sc, r, p);
        wide_fence();
    }
    const Fp2 a = f2_half(m_ld_f2(sc, 0)), b = m_ld_f2(sc, 1), c = m_ld_f2(sc, 2), j = m_ld_f2(sc, 4);
    const Fp2 h = f2_sub(m_ld_f2(sc, 3), f2_add(b, c));                      // 2YZ
    {   // round 2: e = 3b' Z^2 | Tz = b h
        const int r = q & 1;
        const Fp2 p = f2_mul(f2_sel(r == 1, b, b3), f2_sel(r == 1, h, c));
        wide_sync();
        m_st_f2(sc, 5 + r, p);
        wide_fence();
    }
    const Fp2 e = m_ld_f2(sc, 5), tz = m_ld_f2(sc, 6);
    const Fp2 f = f2_add(f2_dbl(e), e);                                      // 9 b' Z^2
    const Fp2 g = f2_half(f2_add(b, f));
    {   // round 3: e^2 | g^2 | a (b - f)
        const int r = q % 3;
        const Fp2 bf = f2_sub(b, f);
        const Fp2 p = f2_mul(f2_sel(r == 0, e, f2_sel(r == 1, g, a)), f2_sel(r == 0, e, f2_sel(r == 1, g, bf)));
        wide_sync();
        m_st_f2(sc, 7 + r, p);
        wide_fence();
    }
    const Fp2 e2 = m_ld_f2(sc, 7), g2 = m_ld_f2(sc, 8), tx = m_ld_f2(sc, 9);
    *l0 = f2_neg(h); *l1 = f2_add(f2_dbl(j), j); *l3 = f2_sub(e, b);
    wide_sync();
    m_st_f2(Tm, 0, tx); m_st_f2(Tm, 1, f2_sub(g2, f2_add(f2_dbl(e2), e2))); m_st_f2(Tm, 2, tz);
    wide_fence();
}
ZKV_W_NI void w_line_add(MRef Tm, MRef sc, const Fp2* qx, const Fp2* qy, Fp2* l0, Fp2* l1, Fp2* l3, int q) {
    const Fp2 x = m_ld_f2(Tm, 0), y = m_ld_f2(Tm, 1), z = m_ld_f2(Tm, 2);
    {   // round 1: qy Z | qx Z
        const int r = q & 1;
        const Fp2 p = f2_mul(f2_sel(r == 1, *qx, *qy), z);
        wide_sync();
        m_st_f2(sc, r, p);
        wide_fence();
    }
    const Fp2 theta = f2_sub(y, m_ld_f2(sc, 0)), lambda = f2_sub(x, m_ld_f2(sc, 1));
    {   // round 2: theta^2 | lambda^2 | theta qx | lambda qy
        const int r = q & 3;
        const Fp2 p = f2_mul(f2_sel((r & 1) == 0, theta, lambda), w_pick4(r, theta, lambda, *qx, *qy));
        wide_sync();
        m_st_f2(sc, 2 + r, p);
        wide_fence();
    }
    const Fp2 c = m_ld_f2(sc, 2), d = m_ld_f2(sc, 3);
    *l3 = f2_sub(m_ld_f2(sc, 4), m_ld_f2(sc, 5));
    {   // round 3: e = lambda d | f = Z c | g = X d
        const int r = q % 3;
        const Fp2 p = f2_mul(f2_sel(r == 0, lambda, f2_sel(r == 1, z, x)), f2_sel(r == 1, c, d));
        wide_sync();
        m_st_f2(sc, 6 + r, p);
        wide_fence();
    }
    const Fp2 e = m_ld_f2(sc, 6), g = m_ld_f2(sc, 8);
    const Fp2 h = f2_sub(f2_add(e, m_ld_f2(sc, 7)), f2_dbl(g));
    {   // round 4: lambda h | theta (g - h) | e Y | Z e
        const int r = q & 3;
        const Fp2 p = f2_mul(w_pick4(r, lambda, theta, e, z), w_pick4(r, h, f2_sub(g, h), y, e));
        wide_sync();
        m_st_f2(sc, 9 + r, p);
        wide_fence();
    }
    *l0 = lambda; *l1 = f2_neg(theta);
    const Fp2 nx = m_ld_f2(sc, 9), ny = f2_sub(m_ld_f2(sc, 10), m_ld_f2(sc, 11)), nz = m_ld_f2(sc, 12);
    wide_sync();
    m_st_f2(Tm, 0, nx); m_st_f2(Tm, 1, ny); m_st_f2(Tm, 2, nz);
    wide_fence();
}

// ---------------------------------------------------------------- Miller loop and final exponentiation on wide slots
// c3 = nl * xs, c4 = c * ys for BOTH fixed pairs of a step in one go (S = 4: one product per slice), then the two sparse products
template <int S> ZKV_HD void fixed_lines_mul_w(MRef fm, MRef sc, const LineAffC& L0, const LineAffC& L1, const G1Norm& n, bool do_l, bool do_c, WL w, MRef red) {
    if (!do_l && !do_c) return;
    const Fp2 v[4] = {f2_const(L0.nl), f2_const(L0.c), f2_const(L1.nl), f2_const(L1.c)};
    const Fp k[4] = {n.lxs, n.lys, n.cxs, n.cys};
    Fp2 c[4];
    w_scale_many<4>(sc, v, k, c, w.q);
    if (do_l) w12_mul_sparse<S>(fm, &c[0], &c[0], &c[1], w, true, red);
    if (do_c) w12_mul_sparse<S>(fm, &c[2], &c[2], &c[3], w, true, red);
}
template <int S> ZKV_HD void var_line_mul_w(MRef fm, MRef sc, const Fp2& l0, const Fp2& l1, const Fp2& l3, const Fp& xs, const Fp& ys, WL w, MRef red) {
    const Fp2 v[2] = {l1, l3};
    const Fp k[2] = {xs, ys};
    Fp2 c[2];
    w_scale_many<2>(sc, v, k, c, w.q);
    w12_mul_sparse<S>(fm, &l0, &c[0], &c[1], w, false, red);
}
// Same schedule as miller_loop_m; the running point T lives in the full-layout slot tm, `sc` is the line functions' scratch, `red` the
// S rows of the slices' partial results (unused for S = 1).
template <int S> ZKV_HD void miller_loop_w(const VkTables& vk, uint32_t flags, const G1Norm& n, const Fp2& bx, const Fp2& by, MRef fm, MRef tm, MRef sc, WL w, MRef red) {
    const int q = w.q;
    const bool do_ab = !(flags & (FL_A_INF | FL_B_INF));
    const bool do_l = !(flags & FL_L_INF) && !vk.skip_fixed[0], do_c = !(flags & FL_C_INF) && !vk.skip_fixed[1];
    w12_set_one(fm, q);
    m_st_f2(tm, 0, bx); m_st_f2(tm, 1, by); m_st_f2(tm, 2, f2_one());
    Fp2 nby = f2_neg(by);
    Fp2 l0, l1, l3;
    int li = 0;
#pragma unroll 1
    for (int i = ZKV_ATE_NAF_LEN - 2; i >= 0; i--) {
        if (i != ZKV_ATE_NAF_LEN - 2) w12_sqr<S>(fm, w, red);
        if (do_ab) {
            w_line_dbl(tm, sc, &l0, &l1, &l3, q);
            var_line_mul_w<S>(fm, sc, l0, l1, l3, n.axs, n.ays, w, red);
        }
        fixed_lines_mul_w<S>(fm, sc, vk.lines[0][li], vk.lines[1][li], n, do_l, do_c, w, red);
        li++;
        int d = ate_naf(i);
        if (d != 0) {
            if (do_ab) {
                Fp2 qy = d > 0 ? by : nby;
                w_line_add(tm, sc, &bx, &qy, &l0, &l1, &l3, q);
                var_line_mul_w<S>(fm, sc, l0, l1, l3, n.axs, n.ays, w, red);
            }
            fixed_lines_mul_w<S>(fm, sc, vk.lines[0][li], vk.lines[1][li], n, do_l, do_c, w, red);
            li++;
        }
    }
    Fp2 qx[2], qy[2];
    g2_frob_affine(qx[0], qy[0], bx, by);
    g2_frob2_affine(qx[1], qy[1], bx, by);
    qy[1] = f2_neg(qy[1]);
#pragma unroll 1
    for (int s = 0; s < 2; s++) {
        if (do_ab) {
            w_line_add(tm, sc, &qx[s], &qy[s], &l0, &l1, &l3, q);
            var_line_mul_w<S>(fm, sc, l0, l1, l3, n.axs, n.ays, w, red);
        }
        fixed_lines_mul_w<S>(fm, sc, vk.lines[0][li], vk.lines[1][li], n, do_l, do_c, w, red);
        li++;
    }
}
// ---------------------------------------------------------------- the Miller loop on TWO wavefronts per proof (k_miller_w64d)
// The running point T and the tangent / chord coefficients do not depend on the accumulator f: one wavefront (the PRODUCER) steps T
// through the 88 line steps and leaves every step's coefficients (l0, l1, l3) in a table in LDS, a second wavefront (the CONSUMER)
// squares f and multiplies the lines in, one proof per wavefront with four slices.  The producer's 3-4 rounds per step were 36 % of the
// one-wavefront kernel's instructions; it runs ahead (fewer instructions per step than the consumer), so the consumer almost never
// waits.  The hand-over is a step counter in LDS: the producer writes the coefficients, fences (work-group scope) and publishes
// `step + 1`; the consumer polls the counter (bounded: a producer that disappeared makes the proof fail, never hang), fences and reads.
// Both wavefronts derive `do_ab` from the same flags, so either both use the table or neither does.
template <class RT> ZKV_HD void miller_lines_producer(const Fp2& bx, const Fp2& by, RT tm, MRef sc, MRef lines, volatile uint32_t* ready, int q) {
    const uint8_t KIND[ZKV_MILLER_STEPS] = ZKV_MILLER_STEP_KIND;
    m_st_f2(tm, 0, bx); m_st_f2(tm, 1, by); m_st_f2(tm, 2, f2_one());
    wide_fence();
    const Fp2 nby = f2_neg(by);
    Fp2 f1x, f1y, f2x, f2y;
    g2_frob_affine(f1x, f1y, bx, by);
    g2_frob2_affine(f2x, f2y, bx, by);
    f2y = f2_neg(f2y);
#pragma unroll 1
    for (int li = 0; li < ZKV_MILLER_STEPS; li++) {
        const int kind = KIND[li];
        Fp2 l0, l1, l3;
        if (kind == 0) w_line_dbl(tm, sc, &l0, &l1, &l3, q);
        else {
            const Fp2 qx = f2_sel(kind == 3, f1x, f2_sel(kind == 4, f2x, bx));
            const Fp2 qy = f2_sel(kind == 1, by, f2_sel(kind == 2, nby, f2_sel(kind == 3, f1y, f2y)));
            w_line_add(tm, sc, &qx, &qy, &l0, &l1, &l3, q);
        }
        const MRef row = m_off(lines, 48 * li);           // three Fp2, full layout; every lane stores its component (same values in every pair)
        m_st_f2(row, 0, l0); m_st_f2(row, 1, l1); m_st_f2(row, 2, l3);
        wide_fence();                                     // the coefficients are in LDS before the counter says so
#if defined(__HIP_DEVICE_COMPILE__)
        if ((threadIdx.x & 63u) == 0) *ready = (uint32_t)(li + 1);
#else
        *ready = (uint32_t)(li + 1);
#endif
    }
}
// false: the producer never published step `need - 1` (cannot happen unless it died); bounded so that nothing can hang the GPU
#if !defined(__HIP_DEVICE_COMPILE__)
void zkv_wide_host_yield();             // host emulation: the producer's threads need the CPU the consumer is spinning on
#endif
ZKV_HD bool miller_lines_wait(volatile uint32_t* ready, uint32_t need) {
#pragma unroll 1
    for (uint32_t spin = 0; spin < (1u << 22); spin++) {
        if (*ready >= need) return true;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_sleep(4);
#else
        zkv_wide_host_yield();
#endif
    }
    return false;
}
// vkp: the context's tables, or nullptr for a single variable pair without fixed pairs (the ecPairing seam)
template <int S> ZKV_HD bool miller_loop_consumer(const VkTables* vkp, uint32_t flags, const G1Norm& n, MRef fm, MRef sc, MRef lines, volatile uint32_t* ready, WL w, MRef red) {
    const uint8_t KIND[ZKV_MILLER_STEPS] = ZKV_MILLER_STEP_KIND;
    const int q = w.q;
    const bool do_ab = !(flags & (FL_A_INF | FL_B_INF));
    const bool do_l = vkp && !(flags & FL_L_INF) && !vkp->skip_fixed[0], do_c = vkp && !(flags & FL_C_INF) && !vkp->skip_fixed[1];
    w12_set_one(fm, q);
    bool ok = true;
#pragma unroll 1
    for (int li = 0; li < ZKV_MILLER_STEPS; li++) {
        if (KIND[li] == 0 && li != 0) w12_sqr<S>(fm, w, red);
        if (do_ab) {
            ok = miller_lines_wait(ready, (uint32_t)(li + 1)) && ok;
            wide_fence();
            const MRef row = m_off(lines, 48 * li);
            const Fp2 l0 = m_ld_f2(row, 0), l1 = m_ld_f2(row, 1), l3 = m_ld_f2(row, 2);
            var_line_mul_w<S>(fm, sc, l0, l1, l3, n.axs, n.ays, w, red);
        }
        if (do_l || do_c) fixed_lines_mul_w<S>(fm, sc, vkp->lines[0][li], vkp->lines[1][li], n, do_l, do_c, w, red);
    }
    return ok;
}

// F <- F^-1, executed redundantly by every pair (all lanes read the whole value, all write the same result).
#if !defined(__HIP_DEVICE_COMPILE__)
uint32_t* zkv_wide_host_pair_tmp();     // host emulation: 96 words shared by the two threads of a pair
#endif
ZKV_HD void f12m_inv_w(MRef F) {
#if defined(__HIP_DEVICE_COMPILE__)
    f12m_inv(F, F);                     // lockstep: every lane has loaded F before the first lane stores
#else
    MRef T = m_ref(zkv_wide_host_pair_tmp() + 8 * zkv_parity(), 1, 16);
    f12m_inv(T, F);
    wide_sync();
    f12m_copy(F, T);
#endif
    wide_fence();
}
// acc <- x^u, same digit schedule as exp_u_m
template <int S> ZKV_HD void exp_u_w(MRef acc, MRef x, MRef W, WL w, MRef red) {
    const int q = w.q;
    const MRef X17 = W, X35 = m_off(W, 96);
    w12_copy(acc, x, q);
#pragma unroll 1
    for (int k = 0; k < 4; k++) w12_cyclo_sqr<S>(acc, w, red);
    w12_mul<S>(X17, acc, x, w, false, red);
    w12_copy(acc, X17, q); w12_cyclo_sqr<S>(acc, w, red);
    w12_mul<S>(X35, acc, x, w, false, red);
    { const int t = u_digit(ZKV_U_DIG_LEN - 1); w12_copy(acc, t == 1 ? x : t == 17 ? X17 : X35, q); }
#pragma unroll 1
    for (int i = ZKV_U_DIG_LEN - 2; i >= 0; i--) {
        w12_cyclo_sqr<S>(acc, w, red);
        const int d = u_digit(i);
        if (d == 0) continue;
        const int m = d < 0 ? -d : d;
        const MRef Sx = m == 1 ? x : m == 17 ? X17 : X35;
        w12_mul<S>(acc, acc, Sx, w, d < 0, red);
    }
}
// Same chain as final_exp_is_one_m.  The inversion and the final comparison run redundantly on every pair.
template <int S> ZKV_HD bool final_exp_is_one_w(MRef F, MRef E, MRef Y1, MRef Y3, MRef Y4, MRef W, MRef acc, WL w, MRef red) {
    const int q = w.q;
    w12_copy(acc, F, q); w12_conj(acc, q);
    f12m_inv_w(F);
    w12_mul<S>(acc, acc, F, w, false, red);          // f^(p^6-1)
    w12_frob(F, acc, 2, q);
    w12_mul<S>(E, F, acc, w, false, red);            // e = ^(p^2+1)
    exp_u_w<S>(acc, E, W, w, red); w12_conj(acc, q);              // y0
    w12_cyclo_sqr<S>(acc, w, red); w12_copy(Y1, acc, q);          // y1
    w12_cyclo_sqr<S>(acc, w, red);                                // y2
    w12_mul<S>(acc, acc, Y1, w, false, red); w12_copy(Y3, acc, q);        // y3
    exp_u_w<S>(acc, Y3, W, w, red); w12_conj(acc, q); w12_copy(Y4, acc, q);   // y4
    w12_cyclo_sqr<S>(acc, w, red); w12_copy(F, acc, q);           // y5
    exp_u_w<S>(acc, F, W, w, red);                                // y6
    w12_conj(Y3, q);
    w12_mul<S>(acc, acc, Y4, w, false, red);                      // y7
    w12_mul<S>(acc, acc, Y3, w, false, red); w12_copy(Y3, acc, q);        // y8
    w12_mul<S>(F, acc, Y1, w, false, red);                        // y9
    w12_mul<S>(acc, acc, Y4, w, false, red);                      // y10
    w12_mul<S>(acc, acc, E, w, false, red);                       // y11
    w12_frob(Y1, F, 1, q);
    w12_mul<S>(acc, Y1, acc, w, false, red);                      // y13
    w12_frob(Y3, Y3, 2, q);
    w12_mul<S>(acc, Y3, acc, w, false, red);                      // y14
    w12_conj(E, q);
    w12_mul<S>(E, E, F, w, false, red);
    w12_frob(E, E, 3, q);                                 // y15
    w12_mul<S>(acc, E, acc, w, false, red);
    return f12m_is_one(acc);
}

}  // namespace zkv
#endif  // ZKV_PAIRED
