// BN254 base field and tower arithmetic for gfx950 kernels.
//
// Replaces what the reference delegates to the EVM precompiles 0x06/0x07/0x08
// (/root/reference/contracts/src/common/groth16.rs:12-14, 60-73, 109-128): Fp Montgomery arithmetic on
// 8 x 32-bit limbs (v_mad_u64_u32 chains), Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3 - (9+u)),
// Fp12 = Fp6[w]/(w^2 - v).  Integer arithmetic only -- no MFMA (not a dense contraction).
//
// The code is plain C++ so tests/ can also compile it for the host (tests/host_sim) and check the
// kernel math without a GPU; the shipped library instantiates it only inside HIP kernels.
#pragma once
#include <stdint.h>
#include "bn254_constants.h"

#if defined(__HIPCC__)
#define ZKV_HD __host__ __device__ __forceinline__
#define ZKV_HD_NI __host__ __device__ __noinline__ inline
#else
#define ZKV_HD inline
#define ZKV_HD_NI inline
#endif

#if defined(ZKV_COUNT_FP_MUL)
static thread_local unsigned long long zkv_fp_mul_counter = 0;      // per thread: the lane-pair host emulation runs two
static thread_local unsigned long long zkv_mad_counter = 0;         // 32 x 32 + 64 multiply-adds (v_mad_u64_u32 on the device) issued by the multipliers
#define ZKV_COUNT_MADS(n) (zkv_mad_counter += (n))
#else
#define ZKV_COUNT_MADS(n) ((void)0)
#endif

// constant tables that are indexed at run time: namespace-scope device constants (a function-local array would be built on the stack)
#if defined(__HIP_DEVICE_COMPILE__)
#define ZKV_TABLE static __device__ const
#else
#define ZKV_TABLE static const
#endif

// Region marks for tools/asm_hist.py (-DZKV_ASM_MARKS builds of the assembly listing only: comments in the instruction stream,
// so that the ISA histogram of a kernel can be split by Fp12-level body).
#if defined(ZKV_ASM_MARKS) && defined(__HIP_DEVICE_COMPILE__)
#define ZKV_MARK(name) asm volatile("; ZKVMARK " name)
#else
#define ZKV_MARK(name) ((void)0)
#endif

// Two wavefronts share a SIMD in the big kernels, and the hardware arbitrates VALU issue between them by priority, then AGE: at equal
// priority the older wavefront runs nearly unimpeded and the younger one gets the leftover slots (measured with per-wavefront clock stamps,
// profiles/round4_*stamps*: in a one-round launch of k_miller2 the wavefronts' own durations spread from 4.8 to 8.1 ms although all do the
// same work).  Once the older one has finished, the younger runs alone -- and a lone wavefront issues at little more than half the rate of two
// -- so every launch ended with a tail of half-idle SIMDs: about 1.3 ms for k_miller2, the "fixed cost per launch" that held a 2^16-proof
// batch 12 % below the 2^20 rate.  zkv_fair_share() makes the two take turns: time is cut into slices of the constant 100 MHz clock, and
// in each slice the wavefront whose slot parity matches takes priority 1 while the other drops to 0; both then finish nearly together
// (2^16 RISC Zero proofs: k_miller2 7.86 -> 7.20 ms, k_finalexp2 4.06 -> 3.64 ms; nothing changes for a 2^20-proof launch, whose SIMDs are
// refilled as wavefronts retire).
#ifndef ZKV_FAIR_SLICE_SHIFT
#define ZKV_FAIR_SLICE_SHIFT 17        /* 2^17 ticks of 10 ns = 1.3 ms: measured best of 2^10 ... 2^20 (profiles/round4_fair_share_slices.txt) */
#endif
ZKV_HD uint32_t zkv_wave_slot_parity() {           // HW_ID.wave_id[0]: the two resident wavefronts of a SIMD sit in slots 0 and 1
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZKV_NO_FAIR_SHARE)
    return __builtin_amdgcn_s_getreg((3 << 11) | 4) & 1u;
#else
    return 0;
#endif
}
ZKV_HD void zkv_fair_share(uint32_t slot_parity) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZKV_NO_FAIR_SHARE)
    const uint32_t t = (uint32_t)(__builtin_amdgcn_s_memrealtime() >> ZKV_FAIR_SLICE_SHIFT);
    if ((t ^ slot_parity) & 1u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#else
    (void)slot_parity;
#endif
}

#include "zkv_modinv.h"

namespace zkv {

struct Fp { uint32_t v[8]; };
// Fp2C: both components, layout-stable (constants, device tables, the one-proof-per-lane kernels).
struct Fp2C { Fp c0, c1; };
#if defined(ZKV_PAIRED)
// Lane-pair mode (k_*2.hip): one proof per PAIR of adjacent lanes; the even lane holds the real component of every
// Fp2 value, the odd lane the imaginary one, and Fp2 products exchange operands with a DPP quad permute.
struct Fp2 { Fp h; };
#else
struct Fp2 { Fp c0, c1; };
#endif
struct Fp6 { Fp2 c0, c1, c2; };

// ---------------------------------------------------------------- carry helpers
// clang lowers __builtin_addc/__builtin_subc chains to v_add_co/v_addc_co (one instruction per limb);
// the arithmetic fallback is for host compilers without the builtins (tests/host_sim with g++).
#if defined(__has_builtin)
#if __has_builtin(__builtin_addc) && __has_builtin(__builtin_subc)
#define ZKV_HAVE_CARRY_BUILTINS 1
#endif
#endif
ZKV_HD uint32_t addc(uint32_t a, uint32_t b, uint32_t& carry) {
#if defined(ZKV_HAVE_CARRY_BUILTINS)
    unsigned co; uint32_t r = __builtin_addc(a, b, carry, &co); carry = co; return r;
#else
    uint64_t t = (uint64_t)a + b + carry;
    carry = (uint32_t)(t >> 32);
    return (uint32_t)t;
#endif
}
ZKV_HD uint32_t subb(uint32_t a, uint32_t b, uint32_t& borrow) {
#if defined(ZKV_HAVE_CARRY_BUILTINS)
    unsigned bo; uint32_t r = __builtin_subc(a, b, borrow, &bo); borrow = bo; return r;
#else
    uint64_t t = (uint64_t)a - b - borrow;
    borrow = (uint32_t)(t >> 32) & 1u;
    return (uint32_t)t;
#endif
}

// ---------------------------------------------------------------- Fp
// Residues in Montgomery form with R = 2^261, stored as 8 x 32-bit limbs, kept in the LOOSE range [0, 2p): every value
// has two representations (x and x + p).  The multipliers then need no final conditional subtraction (their result is
// < V / 2^261 + p < 2p for every column value V < p * 2^261), additions and subtractions reduce against 2p instead of p at
// the same cost, and only comparisons (fp_is_zero, fp_eq) and the conversion out of Montgomery form (fp_to_raw) look at
// both representations.  Constants are canonical, which is a special case.
ZKV_HD Fp fp_zero() { Fp r; for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
ZKV_HD Fp fp_one() { Fp r = ZKV_FP_ONE; return r; }
ZKV_HD bool fp_is_zero(const Fp& a) {          // a in {0, p}
    const uint32_t P[8] = ZKV_FP_P_LIMBS;
    uint32_t o = 0, q = 0;
    for (int i = 0; i < 8; i++) { o |= a.v[i]; q |= a.v[i] ^ P[i]; }
    return o == 0 || q == 0;
}
// raw 256-bit compare a >= m (m given as limbs)
ZKV_HD bool u256_geq(const uint32_t* a, const uint32_t* m) {
    uint32_t br = 0;
    for (int i = 0; i < 8; i++) (void)subb(a[i], m[i], br);
    return br == 0;
}
ZKV_HD Fp fp_add(const Fp& a, const Fp& b) {
    const uint32_t P[8] = ZKV_FP_2P_LIMBS;
    Fp t, s; uint32_t c = 0, br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t.v[i] = addc(a.v[i], b.v[i], c);      // < 4p < 2^256, no carry out
#pragma unroll
    for (int i = 0; i < 8; i++) s.v[i] = subb(t.v[i], P[i], br);
#pragma unroll
    for (int i = 0; i < 8; i++) t.v[i] = br ? t.v[i] : s.v[i];
    return t;
}
ZKV_HD Fp fp_sub(const Fp& a, const Fp& b) {
    const uint32_t P[8] = ZKV_FP_2P_LIMBS;
    Fp t; uint32_t br = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t.v[i] = subb(a.v[i], b.v[i], br);
    uint32_t mask = 0u - br;
#pragma unroll
    for (int i = 0; i < 8; i++) t.v[i] = addc(t.v[i], P[i] & mask, c);
    return t;
}
ZKV_HD bool fp_eq(const Fp& a, const Fp& b) { return fp_is_zero(fp_sub(a, b)); }      // a - b in {0, p} after the loose subtraction
ZKV_HD Fp fp_neg(const Fp& a) { return fp_sub(fp_zero(), a); }
ZKV_HD Fp fp_dbl(const Fp& a) { return fp_add(a, a); }
// a / 2: add p when a is odd (p is odd, so the sum is even and < 3p < 2^256), then shift right by one; result < 2p.
ZKV_HD Fp fp_half(const Fp& a) {
    const uint32_t P[8] = ZKV_FP_P_LIMBS;
    const uint32_t m = 0u - (a.v[0] & 1u);
    Fp t; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t.v[i] = addc(a.v[i], P[i] & m, c);
    Fp r;
#pragma unroll
    for (int i = 0; i < 7; i++) r.v[i] = (t.v[i] >> 1) | (t.v[i + 1] << 31);
    r.v[7] = t.v[7] >> 1;
    return r;
}
// a + b without the modular reduction: for sums that only feed a multiplier (which accepts any 256-bit operand).
// At most TWO loose values may be summed this way: 4p < 2^256 and every product bound in this file assumes operands < 4p.
ZKV_HD Fp fp_add_nr(const Fp& a, const Fp& b) {
    Fp t; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t.v[i] = addc(a.v[i], b.v[i], c);
    return t;
}
// Two independent modular additions / subtractions with their carry chains interleaved limb by limb: gfx950 needs
// wait states between a carry-writing VALU op and the dependent v_addc/v_subb, and the second chain fills them.
ZKV_HD void fp_add_x2(const Fp& a0, const Fp& b0, const Fp& a1, const Fp& b1, Fp& r0, Fp& r1) {
    const uint32_t P[8] = ZKV_FP_2P_LIMBS;
    Fp t0, t1, s0, s1; uint32_t c0 = 0, c1 = 0, w0 = 0, w1 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { t0.v[i] = addc(a0.v[i], b0.v[i], c0); t1.v[i] = addc(a1.v[i], b1.v[i], c1); }
#pragma unroll
    for (int i = 0; i < 8; i++) { s0.v[i] = subb(t0.v[i], P[i], w0); s1.v[i] = subb(t1.v[i], P[i], w1); }
#pragma unroll
    for (int i = 0; i < 8; i++) { r0.v[i] = w0 ? t0.v[i] : s0.v[i]; r1.v[i] = w1 ? t1.v[i] : s1.v[i]; }
}
ZKV_HD void fp_sub_x2(const Fp& a0, const Fp& b0, const Fp& a1, const Fp& b1, Fp& r0, Fp& r1) {
    const uint32_t P[8] = ZKV_FP_2P_LIMBS;
    Fp t0, t1; uint32_t w0 = 0, w1 = 0, c0 = 0, c1 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { t0.v[i] = subb(a0.v[i], b0.v[i], w0); t1.v[i] = subb(a1.v[i], b1.v[i], w1); }
    uint32_t m0 = 0u - w0, m1 = 0u - w1;
#pragma unroll
    for (int i = 0; i < 8; i++) { r0.v[i] = addc(t0.v[i], P[i] & m0, c0); r1.v[i] = addc(t1.v[i], P[i] & m1, c1); }
}

// Montgomery product a*b*2^-261 mod p.
//
// gfx950 has no carry-in on v_mad_u64_u32 and needs wait states between VCC-chained adds, so a 32-bit-limb CIOS
// spends more issue slots on carries, zero-extension moves and s_nops than on multiplies (measured: 590
// instructions, 128 of them multiplies).  The product is therefore formed on 9 x 29-bit limbs: every partial
// product is < 2^58, so a 64-bit column accumulator takes all 18 terms of a column (9 of a*b, 9 of m*p) without
// any carry handling -- one v_mad_u64_u32 per term and nothing else -- and carries are extracted once per column
// with a 64-bit shift.  Operands are unpacked from / packed to the canonical 8 x 32-bit form around the core.
// Columns are signed 64-bit so that differences of products (Fp2 Karatsuba below) reduce with the same routine;
//   |col[k]| <= 9*(2^30)^2 (lazy-sum operands) resp. 18*(2^29)^2 + 9*(2^29)^2 + carry  <  2^63.
ZKV_HD void fp_unpack29(const Fp& a, uint32_t x[9]) {            // 8 x 32 -> 9 x 29 (any 256-bit value)
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int bit = 29 * k, w = bit >> 5, s = bit & 31;
        uint32_t v = a.v[w] >> s;
        if (s > 3 && w < 7) v |= a.v[w + 1] << (32 - s);
        x[k] = v & 0x1fffffffu;
    }
}
template <typename COL>
ZKV_HD void fp_mac81(COL col[18], const uint32_t x[9], const uint32_t y[9]) {   // col += x * y, 81 independent-column MACs
    ZKV_COUNT_MADS(81);
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
        for (int j = 0; j < 9; j++) col[i + j] = (COL)((uint64_t)col[i + j] + (uint64_t)x[i] * y[j]);
    }
}
// Montgomery reduction of an 18-column value V >= 0 (V = sum col[k] 2^(29k)): returns V * 2^-261 mod p in the loose range:
// the result is < V / 2^261 + p, which is < 2p because every caller keeps V < p * 2^261 (about 169 p^2; operands are < 4p).
// The reduction proper: nine 29-bit limbs of V * 2^-261 (limbs 0..7 below 2^29, limb 8 the rest), not yet packed.
template <typename COL>                          // int64_t: signed columns (arithmetic carries); uint64_t: all terms >= 0
ZKV_HD void fp_reduce_cols_limbs(COL col[18], uint32_t (&r)[9]) {
    ZKV_COUNT_MADS(81);
    const uint32_t P29[9] = ZKV_FP_P29_LIMBS;
    const uint32_t M29 = 0x1fffffffu;
#pragma unroll
    for (int i = 0; i < 9; i++) {                 // one 29-bit digit per step
        uint32_t m = ((uint32_t)col[i] * ZKV_FP_INV29) & M29;
#pragma unroll
        for (int j = 0; j < 9; j++) col[i + j] = (COL)((uint64_t)col[i + j] + (uint64_t)m * P29[j]);
        col[i + 1] += col[i] >> 29;               // low 29 bits of col[i] are now zero; arithmetic shift when signed
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
        r[k - 9] = (uint32_t)col[k] & M29;
        col[k + 1] += col[k] >> 29;
    }
    r[8] = (uint32_t)col[17];
}
ZKV_HD Fp fp_pack29(const uint32_t (&r)[9]) {    // 9 x 29 -> 8 x 32 (limbs 0..7 below 2^29, value below 2^256)
    Fp o;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const int bit = 32 * w, k = bit / 29, s = bit - 29 * k, got = 29 - s;
        uint32_t v = r[k] >> s;
        if (k + 1 < 9) v |= r[k + 1] << got;
        if (got + 29 < 32 && k + 2 < 9) v |= r[k + 2] << (got + 29);
        o.v[w] = v;
    }
    return o;
}
template <typename COL>
ZKV_HD Fp fp_reduce_cols(COL col[18]) {
    uint32_t r[9];
    fp_reduce_cols_limbs(col, r);
    return fp_pack29(r);                          // the value is < 2p < 2^255
}
#if defined(ZKV_FP_MUL_NOINLINE)
ZKV_HD_NI
#else
ZKV_HD
#endif
Fp fp_mul(Fp a, Fp b) {   // by value: 16 VGPRs in, 8 out, no scratch traffic at the call
#if defined(ZKV_COUNT_FP_MUL)
    zkv_fp_mul_counter++;     // host-only op counter of tests/host_sim (algorithmic work per stage)
#endif
    uint32_t x[9], y[9];
    fp_unpack29(a, x); fp_unpack29(b, y);
    int64_t col[18];
#pragma unroll
    for (int k = 0; k < 18; k++) col[k] = 0;
    fp_mac81(col, x, y);
    return fp_reduce_cols(col);
}
// a^2: the 36 cross products once, against doubled limbs (45 instead of 81 column terms; one unpack).  Used by the one-value-per-lane
// code (point doublings and additions of k_msm and the PLONK stage); the lane-pair Fp2 squaring multiplies two different values.
#if defined(ZKV_FP_MUL_NOINLINE)
ZKV_HD_NI
#else
ZKV_HD
#endif
Fp fp_sqr(Fp a) {
#if defined(ZKV_COUNT_FP_MUL)
    zkv_fp_mul_counter++;
#endif
    uint32_t x[9], d[9];
    fp_unpack29(a, x);
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = x[i] << 1;            // < 2^30: a column holds at most 4 terms below 2^59 and one below 2^58
    uint64_t col[18];
    ZKV_COUNT_MADS(45);
#pragma unroll
    for (int k = 0; k < 18; k++) col[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        col[2 * i] += (uint64_t)x[i] * x[i];
#pragma unroll
        for (int j = i + 1; j < 9; j++) col[i + j] += (uint64_t)d[i] * x[j];
    }
    return fp_reduce_cols(col);
}

ZKV_HD Fp fp_from_raw(const uint32_t* limbs) {        // canonical value < p -> Montgomery form
    Fp t, r2 = ZKV_FP_R2;
    for (int i = 0; i < 8; i++) t.v[i] = limbs[i];
    return fp_mul(t, r2);
}
ZKV_HD void fp_to_raw(uint32_t* limbs, const Fp& a) {      // out of Montgomery form, canonical
    const uint32_t P[8] = ZKV_FP_P_LIMBS;
    Fp one = fp_zero(); one.v[0] = 1;
    Fp t = fp_mul(a, one), s; uint32_t br = 0;            // t <= p
    for (int i = 0; i < 8; i++) s.v[i] = subb(t.v[i], P[i], br);
    for (int i = 0; i < 8; i++) limbs[i] = br ? t.v[i] : s.v[i];
}
// a^(p-2); inv(0) = 0.  Sliding window of width 2 over a, a^3 (schedule generated by gen_constants.py): a loop, not unrolled.
// Kept as the independent check of fp_inv (tests/host_sim) -- the kernels invert with the division steps of zkv_modinv.h.
ZKV_HD Fp fp_inv_fermat(const Fp& a) {
    const uint8_t S[ZKV_FP_INV_SCHED_LEN] = ZKV_FP_INV_SCHED;
    const Fp a3 = fp_mul(fp_sqr(a), a);
    Fp acc = (S[0] & 7) == 3 ? a3 : a;
#pragma unroll 1
    for (int i = 1; i < ZKV_FP_INV_SCHED_LEN; i++) {
        const int nsq = S[i] >> 3, v = S[i] & 7;
#pragma unroll 1
        for (int k = 0; k < nsq; k++) acc = fp_sqr(acc);
        if (v) {
            Fp m;
#pragma unroll
            for (int k = 0; k < 8; k++) m.v[k] = v == 3 ? a3.v[k] : a.v[k];
            acc = fp_mul(acc, m);
        }
    }
    return acc;
}

// 1/a in Montgomery form; inv(0) = 0.  (aR)^-1 by safegcd division steps (no multiplications, ~1/8 of the Fermat chain's
// instructions), then one multiplication by R^3 brings a^-1 R^-1 back to a^-1 R.  Accepts the loose range.
ZKV_HD Fp fp_inv(const Fp& a) {
    const int32_t M[9] = ZKV_FP_M30_LIMBS;
    const Fp r3 = ZKV_FP_R3;
    Fp t = a;
    modinv30(t.v, M, ZKV_FP_MINV30);
    return fp_mul(t, r3);
}

// N independent modular additions / subtractions, carry chains interleaved limb by limb (same idea as *_x2).
template <int N>
ZKV_HD void fp_add_n(const Fp (&a)[N], const Fp (&b)[N], Fp (&r)[N]) {
    const uint32_t P[8] = ZKV_FP_2P_LIMBS;
    Fp t[N], s[N]; uint32_t c[N], w[N];
#pragma unroll
    for (int k = 0; k < N; k++) { c[k] = 0; w[k] = 0; }
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int k = 0; k < N; k++) t[k].v[i] = addc(a[k].v[i], b[k].v[i], c[k]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int k = 0; k < N; k++) s[k].v[i] = subb(t[k].v[i], P[i], w[k]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int k = 0; k < N; k++) r[k].v[i] = w[k] ? t[k].v[i] : s[k].v[i];
    }
}
template <int N>
ZKV_HD void fp_sub_n(const Fp (&a)[N], const Fp (&b)[N], Fp (&r)[N]) {
    const uint32_t P[8] = ZKV_FP_2P_LIMBS;
    Fp t[N]; uint32_t w[N], c[N], m[N];
#pragma unroll
    for (int k = 0; k < N; k++) { c[k] = 0; w[k] = 0; }
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int k = 0; k < N; k++) t[k].v[i] = subb(a[k].v[i], b[k].v[i], w[k]);
    }
#pragma unroll
    for (int k = 0; k < N; k++) m[k] = 0u - w[k];
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int k = 0; k < N; k++) r[k].v[i] = addc(t[k].v[i], P[i] & m[k], c[k]);
    }
}

// ---------------------------------------------------------------- Fp2
#if !defined(ZKV_PAIRED)
ZKV_HD Fp2 f2_const(const Fp2C& c) { Fp2 r; r.c0 = c.c0; r.c1 = c.c1; return r; }
ZKV_HD Fp2 f2_zero() { Fp2 r; r.c0 = fp_zero(); r.c1 = fp_zero(); return r; }
ZKV_HD Fp2 f2_one() { Fp2 r; r.c0 = fp_one(); r.c1 = fp_zero(); return r; }
ZKV_HD bool f2_is_zero(const Fp2& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
ZKV_HD bool f2_eq(const Fp2& a, const Fp2& b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
ZKV_HD Fp2 f2_add(const Fp2& a, const Fp2& b) { Fp2 r; fp_add_x2(a.c0, b.c0, a.c1, b.c1, r.c0, r.c1); return r; }
ZKV_HD Fp2 f2_sub(const Fp2& a, const Fp2& b) { Fp2 r; fp_sub_x2(a.c0, b.c0, a.c1, b.c1, r.c0, r.c1); return r; }
ZKV_HD Fp2 f2_neg(const Fp2& a) { Fp2 r; Fp z = fp_zero(); fp_sub_x2(z, a.c0, z, a.c1, r.c0, r.c1); return r; }
// lazy sum (components < 2p) that may only be passed to f2_mul
ZKV_HD Fp2 f2_add_nr(const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = fp_add_nr(a.c0, b.c0); r.c1 = fp_add_nr(a.c1, b.c1); return r; }
ZKV_HD Fp2 f2_dbl(const Fp2& a) { return f2_add(a, a); }
ZKV_HD Fp2 f2_half(const Fp2& a) { Fp2 r; r.c0 = fp_half(a.c0); r.c1 = fp_half(a.c1); return r; }
ZKV_HD Fp2 f2_conj(const Fp2& a) { Fp2 r; r.c0 = a.c0; r.c1 = fp_neg(a.c1); return r; }
// Fp2 product with the Montgomery reductions shared: three 81-term column products (a0 b0, a1 b1,
// (a0+a1)(b0+b1)) and only TWO reductions -- c1 = M - T0 - T1 is non-negative column by column, c0 = T0 - T1 + 16 p^2
// is non-negative as a whole and reduces on signed columns.  405 multiplies instead of 486.
// Operand components may be lazy sums < 4p (so T1 < 16 p^2 and every total stays < 64 p^2 < 169 p^2).
ZKV_HD void f2_mul_core(const Fp& a0, const Fp& a1, const Fp& b0, const Fp& b1, Fp& c0, Fp& c1) {
#if defined(ZKV_COUNT_FP_MUL)
    zkv_fp_mul_counter += 3;
#endif
    const uint64_t C16P2[18] = ZKV_FP_16P2_COLS;
    uint32_t x0[9], x1[9], y0[9], y1[9], sx[9], sy[9];
    fp_unpack29(a0, x0); fp_unpack29(a1, x1); fp_unpack29(b0, y0); fp_unpack29(b1, y1);
#pragma unroll
    for (int i = 0; i < 9; i++) { sx[i] = x0[i] + x1[i]; sy[i] = y0[i] + y1[i]; }
    int64_t t0[18], t1[18], m[18];
#pragma unroll
    for (int k = 0; k < 18; k++) { t0[k] = 0; t1[k] = 0; m[k] = 0; }
    fp_mac81(t0, x0, y0); fp_mac81(t1, x1, y1); fp_mac81(m, sx, sy);
#pragma unroll
    for (int k = 0; k < 18; k++) {
        m[k] = (int64_t)((uint64_t)m[k] - (uint64_t)t0[k] - (uint64_t)t1[k]);
        t0[k] = (int64_t)((uint64_t)t0[k] - (uint64_t)t1[k] + C16P2[k]);
    }
    c0 = fp_reduce_cols(t0);
    c1 = fp_reduce_cols(m);
}
#if defined(__HIP_DEVICE_COMPILE__) && defined(ZKV_FP_MUL_NOINLINE)
// Non-inlined on the device with all 32 operand limbs as scalar parameters: the AMDGPU calling convention passes
// them in v0..v31 and returns the 16 result limbs in v0..v15 -- no stack traffic at the call.
#define ZKV_L8(p) uint32_t p##0, uint32_t p##1, uint32_t p##2, uint32_t p##3, uint32_t p##4, uint32_t p##5, uint32_t p##6, uint32_t p##7
#define ZKV_A8(f) f.v[0], f.v[1], f.v[2], f.v[3], f.v[4], f.v[5], f.v[6], f.v[7]
#define ZKV_MK(f, p) Fp f; f.v[0] = p##0; f.v[1] = p##1; f.v[2] = p##2; f.v[3] = p##3; f.v[4] = p##4; f.v[5] = p##5; f.v[6] = p##6; f.v[7] = p##7
__device__ __noinline__ inline Fp2 f2_mul_ni(ZKV_L8(pa), ZKV_L8(pb), ZKV_L8(pc), ZKV_L8(pd)) {
    ZKV_MK(a0, pa); ZKV_MK(a1, pb); ZKV_MK(b0, pc); ZKV_MK(b1, pd);
    Fp2 r; f2_mul_core(a0, a1, b0, b1, r.c0, r.c1);
    return r;
}
ZKV_HD Fp2 f2_mul(const Fp2& a, const Fp2& b) { return f2_mul_ni(ZKV_A8(a.c0), ZKV_A8(a.c1), ZKV_A8(b.c0), ZKV_A8(b.c1)); }
#else
ZKV_HD Fp2 f2_mul(const Fp2& a, const Fp2& b) { Fp2 r; f2_mul_core(a.c0, a.c1, b.c0, b.c1, r.c0, r.c1); return r; }
#endif
ZKV_HD Fp2 f2_sqr(const Fp2& a) {           // reduced (< 2p) input
    Fp2 r;
    r.c0 = fp_mul(fp_add_nr(a.c0, a.c1), fp_sub(a.c0, a.c1));
    r.c1 = fp_mul(fp_add_nr(a.c0, a.c0), a.c1);
    return r;
}
ZKV_HD Fp2 f2_mul_fp(const Fp2& a, const Fp& k) { Fp2 r; r.c0 = fp_mul(a.c0, k); r.c1 = fp_mul(a.c1, k); return r; }
ZKV_HD Fp2 f2_mul_xi(const Fp2& a) {          // (9+u)(a0 + a1 u) = (9a0 - a1) + (9a1 + a0) u
    Fp t0 = fp_dbl(fp_dbl(fp_dbl(a.c0))), t1 = fp_dbl(fp_dbl(fp_dbl(a.c1)));
    Fp2 r; r.c0 = fp_sub(fp_add(t0, a.c0), a.c1); r.c1 = fp_add(fp_add(t1, a.c1), a.c0);
    return r;
}
ZKV_HD Fp2 f2_inv(const Fp2& a) {
    Fp n = fp_add(fp_sqr(a.c0), fp_sqr(a.c1));
    Fp i = fp_inv(n);
    Fp2 r; r.c0 = fp_mul(a.c0, i); r.c1 = fp_neg(fp_mul(a.c1, i));
    return r;
}

#else   // ---------------------------------------------------------------- Fp2, lane-pair mode
#if defined(__HIP_DEVICE_COMPILE__)
// parity of the lane within its wavefront (= parity of threadIdx.x: blocks are whole wavefronts), from the lane counter: a non-inlined
// leaf that reads threadIdx is handed the packed work-item id in v31.
ZKV_HD uint32_t zkv_parity() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & 1u; }
// value held by the other lane of the pair: v_mov_b32_dpp quad_perm:[1,0,3,2]
ZKV_HD uint32_t zkv_partner_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true); }
// the even (odd) lane's value in both lanes of the pair: quad_perm:[0,0,2,2] ([1,1,3,3])
ZKV_HD uint32_t zkv_pair_even_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xA0, 0xF, 0xF, true); }
ZKV_HD uint32_t zkv_pair_odd_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xF5, 0xF, 0xF, true); }
#else
uint32_t zkv_parity();                      // host emulation of a lane pair (tests/host_sim, two threads)
uint32_t zkv_partner_u32(uint32_t x);
inline uint32_t zkv_pair_even_u32(uint32_t x) { const uint32_t o = zkv_partner_u32(x); return zkv_parity() ? o : x; }
inline uint32_t zkv_pair_odd_u32(uint32_t x) { const uint32_t o = zkv_partner_u32(x); return zkv_parity() ? x : o; }
#endif
ZKV_HD Fp zkv_partner(const Fp& a) {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = zkv_partner_u32(a.v[i]);
    return r;
}
ZKV_HD Fp fp_sel(bool odd, const Fp& if_odd, const Fp& if_even) {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = odd ? if_odd.v[i] : if_even.v[i];
    return r;
}
ZKV_HD Fp2 f2_const(const Fp2C& c) { Fp2 r; r.h = fp_sel(zkv_parity() != 0, c.c1, c.c0); return r; }
ZKV_HD Fp2 f2_zero() { Fp2 r; r.h = fp_zero(); return r; }
ZKV_HD Fp2 f2_one() { Fp2 r; r.h = fp_sel(zkv_parity() != 0, fp_zero(), fp_one()); return r; }
ZKV_HD bool f2_is_zero(const Fp2& a) {
    uint32_t o = fp_is_zero(a.h) ? 0u : 1u;
    o |= zkv_partner_u32(o);
    return o == 0;
}
ZKV_HD bool f2_eq(const Fp2& a, const Fp2& b) {
    uint32_t o = fp_eq(a.h, b.h) ? 0u : 1u;
    o |= zkv_partner_u32(o);
    return o == 0;
}
ZKV_HD Fp2 f2_add(const Fp2& a, const Fp2& b) { Fp2 r; r.h = fp_add(a.h, b.h); return r; }
ZKV_HD Fp2 f2_sub(const Fp2& a, const Fp2& b) { Fp2 r; r.h = fp_sub(a.h, b.h); return r; }
ZKV_HD Fp2 f2_neg(const Fp2& a) { Fp2 r; r.h = fp_neg(a.h); return r; }
ZKV_HD Fp2 f2_add_nr(const Fp2& a, const Fp2& b) { Fp2 r; r.h = fp_add_nr(a.h, b.h); return r; }
ZKV_HD Fp2 f2_dbl(const Fp2& a) { return f2_add(a, a); }
ZKV_HD Fp2 f2_half(const Fp2& a) { Fp2 r; r.h = fp_half(a.h); return r; }
ZKV_HD Fp2 f2_conj(const Fp2& a) { Fp2 r; r.h = fp_sel(zkv_parity() != 0, fp_neg(a.h), a.h); return r; }
// (a0 + a1 u)(b0 + b1 u): the even lane forms a0 b0 + a1 (8p - b1), the odd lane a0 b1 + a1 b0 -- two 81-term column
// products and ONE Montgomery reduction per lane, all terms non-negative (8p - b1 in borrow-free 29-bit limbs).
// Components may be lazy sums < 4p: the column value stays below 16 p^2 + 32 p^2.
#if defined(ZKV_FP_MUL_NOINLINE)
ZKV_HD_NI
#else
ZKV_HD
#endif
Fp f2_mul_lane(Fp my_a, Fp my_b) {
#if defined(ZKV_COUNT_FP_MUL)
    zkv_fp_mul_counter += 2;
#endif
    const uint32_t FAT[9] = ZKV_FP_FAT8P_LIMBS;
    const bool odd = zkv_parity() != 0;
    uint32_t xa[9], xo[9], yb[9], U[9], V[9];
    fp_unpack29(my_a, xa); fp_unpack29(my_b, yb);        // each lane unpacks its own operands once ...
    // ... and the limbs travel: a is swapped; of b both lanes take the even lane's b0 (for the product with their own a) and the odd
    // lane's b1 (for the product with the partner's a), which the even lane turns into 8p - b1.  Three DPP moves per limb and no
    // selects (the swap-and-select form cost 18 more v_mov).
#pragma unroll
    for (int i = 0; i < 9; i++) { xo[i] = zkv_partner_u32(xa[i]); U[i] = zkv_pair_even_u32(yb[i]); V[i] = zkv_pair_odd_u32(yb[i]); }
    if (!odd) {
#pragma unroll
        for (int i = 0; i < 9; i++) V[i] = FAT[i] - V[i];
    }
    uint64_t col[18];
#pragma unroll
    for (int k = 0; k < 18; k++) col[k] = 0;
    fp_mac81(col, xa, U); fp_mac81(col, xo, V);
    return fp_reduce_cols(col);
}
ZKV_HD Fp2 f2_mul(const Fp2& a, const Fp2& b) { Fp2 r; r.h = f2_mul_lane(a.h, b.h); return r; }
// a b + c d with ONE Montgomery reduction per lane: four 81-term column products (even lane a0 b0 + a1 (8p - b1) + c0 d0 + c1 (8p - d1),
// odd lane a0 b1 + a1 b0 + c0 d1 + c1 d0).  All four operands must be reduced values (< 2p: limbs < 2^29 after unpacking; no lazy
// sums here): a column then holds at most 18 products below 2^58, 18 below 2^59 (the 8p - x limbs are below 2^30) and the 9
// reduction terms below 2^58 -- 63 * 2^58 < 2^64 -- and the value stays below 40 p^2 < 169 p^2.  Used where two products are only
// ever added (the sparse fixed-line products of the Miller loop): 405 multiplies and one reduce / pack set instead of 486 and two.
// (History: a leaf taking four Fp by value had two of them passed through scratch memory -- the ABI grants aggregates 16 argument
// registers -- 8.6 GB per 2^20-proof launch; a leaf reading two operands from LDS fixed that; the form below replaced both.)
// Operands arrive unpacked and exchanged, for callers that use every operand several times:
//   multiplicand form (f2_limbs_x): this lane's nine limbs and the partner's;
//   multiplier form   (f2_limbs_y): U = the even lane's component in both lanes, V = the odd lane's (8p - it in the even lane).
ZKV_HD void f2_limbs_x(const Fp& v, uint32_t (&own)[9], uint32_t (&par)[9]) {
    fp_unpack29(v, own);
#pragma unroll
    for (int i = 0; i < 9; i++) par[i] = zkv_partner_u32(own[i]);
}
ZKV_HD void f2_limbs_y(const Fp& v, uint32_t (&U)[9], uint32_t (&V)[9]) {
    const uint32_t FAT[9] = ZKV_FP_FAT8P_LIMBS;
    uint32_t y[9];
    fp_unpack29(v, y);
#pragma unroll
    for (int i = 0; i < 9; i++) { U[i] = zkv_pair_even_u32(y[i]); V[i] = zkv_pair_odd_u32(y[i]); }
    if (zkv_parity() == 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) V[i] = FAT[i] - V[i];
    }
}
ZKV_HD Fp f2_dot2_limbs(const uint32_t (&ao)[9], const uint32_t (&ap)[9], const uint32_t (&bU)[9], const uint32_t (&bV)[9],
                        const uint32_t (&co)[9], const uint32_t (&cp)[9], const uint32_t (&dU)[9], const uint32_t (&dV)[9]) {
#if defined(ZKV_COUNT_FP_MUL)
    zkv_fp_mul_counter += 4;
#endif
    uint64_t col[18];
#pragma unroll
    for (int k = 0; k < 18; k++) col[k] = 0;
    fp_mac81(col, ao, bU); fp_mac81(col, ap, bV); fp_mac81(col, co, dU); fp_mac81(col, cp, dV);
    return fp_reduce_cols(col);
}
// one product from prepared operands (what f2_mul_lane computes; the multiplicand and multiplier may be lazy sums below 4p)
ZKV_HD Fp f2_mul_limbs(const uint32_t (&ao)[9], const uint32_t (&ap)[9], const uint32_t (&bU)[9], const uint32_t (&bV)[9]) {
#if defined(ZKV_COUNT_FP_MUL)
    zkv_fp_mul_counter += 2;
#endif
    uint64_t col[18];
#pragma unroll
    for (int k = 0; k < 18; k++) col[k] = 0;
    fp_mac81(col, ao, bU); fp_mac81(col, ap, bV);
    return fp_reduce_cols(col);
}
// ---------------------------------------------------------------- resident 29-bit limbs (L9), lane-pair kernels
// The accumulator of the final exponentiation stays UNPACKED between Fp12 operations: a value is nine 29-bit limbs in nine words
// (limbs 0..7 below 2^29, value below 2p), which is what the column multiplier consumes and produces.  That removes the unpack /
// pack around every product (54 of the 160 non-multiply instructions of a lane product) and, more importantly, the carry chains: on
// gfx950 every v_addc / v_subb needs a wait state after the instruction that wrote VCC (the ISA listing of the packed cyclotomic
// squaring holds 808 carry instructions and 704 s_nop), whereas limb-wise sums have no carries at all.  Linear combinations of
// several values -- Karatsuba's differences, 3t +- 2z of the cyclotomic squaring, the xi-multiplication 9x -+ x' -- are formed in
// ONE pass per result (l9_lincomb): per limb one 32 x 32 + 64 multiply-add per term with a small constant, one more with the
// quotient estimate times p, and a carry step; the result is normalised and below 1.01 p.
struct L9 { uint32_t l[9]; };
ZKV_HD L9 l9_from_fp(const Fp& a) { L9 r; fp_unpack29(a, r.l); return r; }
ZKV_HD Fp l9_to_fp(const L9& a) { return fp_pack29(a.l); }      // normalised limbs, value < 2^256
ZKV_HD L9 l9_partner(const L9& a) {
    L9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = zkv_partner_u32(a.l[i]);
    return r;
}
// (a0 + a1 u)(b0 + b1 u), this lane's component: what f2_mul_lane computes, on limbs and without the unpack / pack.  a, b: this lane's
// components.  The multiplicand's limbs may be a lazy sum below 2^30; the multiplier's are normalised (below 2^29) with a value below 4p
// (8p - b1 is formed limb-wise).  The limbs travel as in f2_mul_lane: a is swapped; of b both lanes take the even lane's b0 and the odd
// lane's b1, which the even lane turns into 8p - b1.  A column holds at most 9 (2 + 4) + 9 = 63 units of 2^58; values: a0 b0 + a1 (8p - b1)
// must stay below 169 p^2.
ZKV_HD L9 l9_mul_core(const L9& a, const L9& b) {
#if defined(ZKV_COUNT_FP_MUL)
    zkv_fp_mul_counter += 2;
#endif
    const uint32_t FAT[9] = ZKV_FP_FAT8P_LIMBS;
    uint32_t xo[9], U[9], V[9];
    uint64_t col[18];
#pragma unroll
    for (int k = 0; k < 18; k++) col[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) U[i] = zkv_pair_even_u32(b.l[i]);
    fp_mac81(col, a.l, U);
#pragma unroll
    for (int i = 0; i < 9; i++) { xo[i] = zkv_partner_u32(a.l[i]); V[i] = zkv_pair_odd_u32(b.l[i]); }
    if (zkv_parity() == 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) V[i] = FAT[i] - V[i];
    }
    fp_mac81(col, xo, V);
    L9 r; fp_reduce_cols_limbs(col, r.l);
    return r;
}
#if defined(__HIP_DEVICE_COMPILE__) && defined(ZKV_FP_MUL_NOINLINE)
// Non-inlined leaf with the 18 operand limbs as scalar parameters (v0..v17 in, v0..v8 out: no stack traffic at the call); the callers
// are the inlined Fp12 bodies of the final exponentiation, which would otherwise hold 18 copies of the product.
#define ZKV_L9P(p) uint32_t p##0, uint32_t p##1, uint32_t p##2, uint32_t p##3, uint32_t p##4, uint32_t p##5, uint32_t p##6, uint32_t p##7, uint32_t p##8
#define ZKV_L9A(f) f.l[0], f.l[1], f.l[2], f.l[3], f.l[4], f.l[5], f.l[6], f.l[7], f.l[8]
#define ZKV_L9MK(f, p) L9 f; f.l[0] = p##0; f.l[1] = p##1; f.l[2] = p##2; f.l[3] = p##3; f.l[4] = p##4; f.l[5] = p##5; f.l[6] = p##6; f.l[7] = p##7; f.l[8] = p##8
__device__ __noinline__ inline L9 l9_mul_ni(ZKV_L9P(pa), ZKV_L9P(pb)) {
    ZKV_L9MK(a, pa); ZKV_L9MK(b, pb);
    return l9_mul_core(a, b);
}
ZKV_HD L9 l9_mul(const L9& a, const L9& b) { return l9_mul_ni(ZKV_L9A(a), ZKV_L9A(b)); }
#else
ZKV_HD L9 l9_mul(const L9& a, const L9& b) { return l9_mul_core(a, b); }
#endif
// sum over j of k_j x_j  (mod p), normalised and below 1.01 p.  x_j: nine signed 32-bit limbs each (normalised values, or lazy limb-wise
// sums / differences of a few of them: |limb| < 2^31); k_j: small integers, possibly different in the two lanes of a pair.
// c: any integer >= 1 + sum over the terms that can be negative of |k_j| * (bound of x_j in units of p): it keeps the quotient
// estimate non-negative.  Let D = floor(p / 2^232) + 1.  The sum X satisfies (T - G) 2^232 < X < (T + G) 2^232 for T = sum k_j x_j[8]
// (every x_j is its top limb times 2^232 plus a lower part below rho_j 2^232, and G = 256 exceeds sum |k_j| rho_j + c by construction), so
// with q = floor((T + c D - c - G) / D) - c:  0 <= X - q p < p + (2 G + c + q + 2) 2^232 < 1.01 p.  The division is one v_mul_hi: with
// M = ceil(2^53 / D) the product (t M) >> 53 equals floor(t / D) for every t < 2^31 (the excess t / 2^53 stays below 1 / D).
struct LTerm { const uint32_t* x; int32_t k; };
#define ZKV_L9_G 256
template <int N> ZKV_HD L9 l9_lincomb(const LTerm (&t)[N], const int c) {
    const uint32_t P29[9] = ZKV_FP_P29_LIMBS;
    const int32_t D = (int32_t)P29[8] + 1;
    const uint32_t M = (uint32_t)(((1ull << 53) + (uint64_t)D - 1) / (uint64_t)D);
    int32_t top = 0;
#pragma unroll
    for (int j = 0; j < N; j++) top += t[j].k * (int32_t)t[j].x[8];
    const uint32_t tt = (uint32_t)(top + c * D - c - ZKV_L9_G);
    const int32_t nq = c - (int32_t)(uint32_t)(((uint64_t)tt * M) >> 53);          // -q
    L9 y; int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int64_t s = carry;
#pragma unroll
        for (int j = 0; j < N; j++) s += (int64_t)t[j].k * (int64_t)(int32_t)t[j].x[i];
        s += (int64_t)nq * (int64_t)(int32_t)P29[i];
        if (i < 8) { y.l[i] = (uint32_t)s & 0x1fffffffu; carry = s >> 29; }
        else y.l[i] = (uint32_t)s;
    }
    return y;
}
ZKV_HD Fp2 f2_sqr(const Fp2& a) {           // reduced (< 2p) input: even lane (a0+a1)(a0-a1), odd lane (2 a1) a0
    const bool odd = zkv_parity() != 0;
    Fp o = zkv_partner(a.h);
    Fp x = fp_add_nr(a.h, fp_sel(odd, a.h, o));
    Fp y = fp_sel(odd, o, fp_sub(a.h, o));
    Fp2 r; r.h = fp_mul(x, y);
    return r;
}
ZKV_HD Fp2 f2_mul_fp(const Fp2& a, const Fp& k) { Fp2 r; r.h = fp_mul(a.h, k); return r; }
// (9+u)(a0 + a1 u) = (9a0 - a1) + (9a1 + a0) u: each lane needs 9 * mine -/+ partner's.  Measured (zkv_diag_mulmod_rate kind 4) the
// round-1 form -- three modular doublings, an addition, a negation, a select and another addition, every one a carry chain with
// its conditional subtraction -- cost 0.69 of an fp_mul, 17-31 % of the time of an Fp12 routine.  Now: the odd lane hands its
// partner 2p - a1 (so that both lanes ADD), T = (mine << 3) + mine + partner's is formed once on nine limbs (T < 20p), a quotient
// estimate from the top 14 bits (one v_mul_hi) takes T below 1.0034 * 2p (checked at every multiple of 2p +- 3 and on random values),
// and a single conditional subtraction restores the loose range [0, 2p).
ZKV_HD Fp2 f2_mul_xi(const Fp2& a) {
    const uint32_t P2[8] = ZKV_FP_2P_LIMBS;
    const bool odd = zkv_parity() != 0;
    Fp n; uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) n.v[i] = subb(P2[i], a.h.v[i], br);            // 2p - mine, in (0, 2p]
    const Fp o = zkv_partner(fp_sel(odd, n, a.h));
    uint32_t t[9], c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = addc(a.h.v[i], o.v[i], c);                // mine + partner's, nine limbs
    t[8] = c;
    uint32_t sh[9];
    sh[0] = a.h.v[0] << 3;
#pragma unroll
    for (int i = 1; i < 8; i++) sh[i] = (a.h.v[i] << 3) | (a.h.v[i - 1] >> 29);
    sh[8] = a.h.v[7] >> 29;
    c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = addc(t[i], sh[i], c);                     // T = 9 mine + partner's < 20p < 2^259
    const uint32_t x = (t[8] << 12) | (t[7] >> 20);                              // floor(T / 2^244)
    const uint32_t q = (uint32_t)(((uint64_t)x * 0x2a4effu) >> 32);              // floor(T / 2p) or one less: 0x2a4eff = floor(2^32 / ceil(2p / 2^244))
    uint64_t carry = 0;
    Fp r, s2; br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { carry += (uint64_t)q * P2[i]; r.v[i] = subb(t[i], (uint32_t)carry, br); carry >>= 32; }       // T - q 2p < 4p: eight limbs
    br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s2.v[i] = subb(r.v[i], P2[i], br);
    Fp2 out;
#pragma unroll
    for (int i = 0; i < 8; i++) out.h.v[i] = br ? r.v[i] : s2.v[i];
    return out;
}
ZKV_HD Fp2 f2_inv(const Fp2& a) {
    Fp sq = fp_sqr(a.h);
    Fp n = fp_add(sq, zkv_partner(sq));
    Fp i = fp_inv(n);
    Fp m = fp_mul(a.h, i);
    Fp2 r; r.h = fp_sel(zkv_parity() != 0, fp_neg(m), m);
    return r;
}
#endif  // ZKV_PAIRED

// ---------------------------------------------------------------- Fp6
ZKV_HD Fp6 f6_zero() { Fp6 r; r.c0 = f2_zero(); r.c1 = f2_zero(); r.c2 = f2_zero(); return r; }
// Fp6 as a flat array of Fp components (3 per lane in lane-pair mode, 6 otherwise) for the interleaved chains.
#if defined(ZKV_PAIRED)
constexpr int F6N = 3;
ZKV_HD void f6_flat(const Fp6& a, Fp (&o)[3]) { o[0] = a.c0.h; o[1] = a.c1.h; o[2] = a.c2.h; }
ZKV_HD Fp6 f6_unflat(const Fp (&o)[3]) { Fp6 r; r.c0.h = o[0]; r.c1.h = o[1]; r.c2.h = o[2]; return r; }
#else
constexpr int F6N = 6;
ZKV_HD void f6_flat(const Fp6& a, Fp (&o)[6]) { o[0] = a.c0.c0; o[1] = a.c0.c1; o[2] = a.c1.c0; o[3] = a.c1.c1; o[4] = a.c2.c0; o[5] = a.c2.c1; }
ZKV_HD Fp6 f6_unflat(const Fp (&o)[6]) { Fp6 r; r.c0.c0 = o[0]; r.c0.c1 = o[1]; r.c1.c0 = o[2]; r.c1.c1 = o[3]; r.c2.c0 = o[4]; r.c2.c1 = o[5]; return r; }
#endif
ZKV_HD Fp6 f6_add(const Fp6& a, const Fp6& b) { Fp x[F6N], y[F6N], r[F6N]; f6_flat(a, x); f6_flat(b, y); fp_add_n<F6N>(x, y, r); return f6_unflat(r); }
ZKV_HD Fp6 f6_sub(const Fp6& a, const Fp6& b) { Fp x[F6N], y[F6N], r[F6N]; f6_flat(a, x); f6_flat(b, y); fp_sub_n<F6N>(x, y, r); return f6_unflat(r); }
ZKV_HD Fp6 f6_neg(const Fp6& a) { Fp x[F6N], z[F6N], r[F6N]; f6_flat(a, x); for (int k = 0; k < F6N; k++) z[k] = fp_zero(); fp_sub_n<F6N>(z, x, r); return f6_unflat(r); }
ZKV_HD Fp6 f6_mul_v(const Fp6& a) { Fp6 r; r.c0 = f2_mul_xi(a.c2); r.c1 = a.c0; r.c2 = a.c1; return r; }
ZKV_HD Fp6 f6_mul(const Fp6& a, const Fp6& b) {
    Fp2 v0 = f2_mul(a.c0, b.c0), v1 = f2_mul(a.c1, b.c1), v2 = f2_mul(a.c2, b.c2);
    Fp6 t, u, w;
    t.c0 = f2_mul(f2_add_nr(a.c1, a.c2), f2_add_nr(b.c1, b.c2));
    t.c1 = f2_mul(f2_add_nr(a.c0, a.c1), f2_add_nr(b.c0, b.c1));
    t.c2 = f2_mul(f2_add_nr(a.c0, a.c2), f2_add_nr(b.c0, b.c2));
    u.c0 = v1; u.c1 = v0; u.c2 = v0;
    w.c0 = v2; w.c1 = v1; w.c2 = v2;
    t = f6_sub(f6_sub(t, u), w);                         // the three Karatsuba cross terms, chains interleaved
    u.c0 = v0; u.c1 = t.c1; u.c2 = t.c2;
    w.c0 = f2_mul_xi(t.c0); w.c1 = f2_mul_xi(v2); w.c2 = v1;
    return f6_add(u, w);
}
// a * (b0 + b1 v)   (5 Fp2 products)
ZKV_HD Fp6 f6_mul_by_01(const Fp6& a, const Fp2& b0, const Fp2& b1) {
    Fp2 v0 = f2_mul(a.c0, b0), v1 = f2_mul(a.c1, b1);
    Fp6 r;
    r.c1 = f2_sub(f2_sub(f2_mul(f2_add_nr(a.c0, a.c1), f2_add_nr(b0, b1)), v0), v1);
    r.c0 = f2_add(v0, f2_mul_xi(f2_mul(a.c2, b1)));
    r.c2 = f2_add(v1, f2_mul(a.c2, b0));
    return r;
}
ZKV_HD Fp6 f6_mul_fp2(const Fp6& a, const Fp2& k) { Fp6 r; r.c0 = f2_mul(a.c0, k); r.c1 = f2_mul(a.c1, k); r.c2 = f2_mul(a.c2, k); return r; }
ZKV_HD Fp6 f6_inv(const Fp6& a) {
    Fp2 A = f2_sub(f2_sqr(a.c0), f2_mul_xi(f2_mul(a.c1, a.c2)));
    Fp2 B = f2_sub(f2_mul_xi(f2_sqr(a.c2)), f2_mul(a.c0, a.c1));
    Fp2 C = f2_sub(f2_sqr(a.c1), f2_mul(a.c0, a.c2));
    Fp2 F = f2_add(f2_mul_xi(f2_add(f2_mul(a.c2, B), f2_mul(a.c1, C))), f2_mul(a.c0, A));
    F = f2_inv(F);
    Fp6 r; r.c0 = f2_mul(A, F); r.c1 = f2_mul(B, F); r.c2 = f2_mul(C, F);
    return r;
}

}  // namespace zkv
