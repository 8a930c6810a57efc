// Modular inversion by the Bernstein-Yang "safegcd" division steps (D. J. Bernstein, B.-Y. Yang, "Fast constant-time gcd computation
// and modular inversion", TCHES 2019), in the half-delta form with batches of 30 steps on signed 30-bit limbs: 600 steps cover every
// 256-bit input for a 256-bit odd modulus (the proven bound is 590).  Why here: the division steps look only at the low 30 bits of
// f and g, so a batch is 30 x ~17 plain 32-bit VALU instructions with no multiplication at all, and every lane of a wavefront runs
// the same straight-line code (masks, no branches).  A Fermat inversion is 252 squarings + 77 multiplications = 329 x 280 instructions;
// this is 20 batches + 20 x two small matrix-vector updates (v_mad_i64_i32), about an eighth of that.
//
// Replaces nothing in the reference by itself: inversions arise inside the arithmetic the reference leaves to the precompiles
// (/root/reference/contracts/src/common/groth16.rs:12-14): affine normalisation after compute_vk_x, the Fp12 inversion of the final
// exponentiation, affine results of the ecAdd / ecMul seam, and the scalar-field inversion of the PLONK verifier.
#pragma once
#include <stdint.h>

namespace zkv {

struct ModInv30Trans { int32_t u, v, q, r; };

// 30 division steps on the low limbs.  zeta = -(delta + 1/2); returns the new zeta and the transition matrix t with
// t * [f, g] = 2^30 * [f', g'].
ZKV_HD int32_t modinv30_divsteps(int32_t zeta, uint32_t f0, uint32_t g0, ModInv30Trans& t) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 6
    for (int i = 0; i < 30; i++) {
        uint32_t m1 = (uint32_t)(zeta >> 31);                 // zeta < 0
        const uint32_t m2 = 0u - (g & 1u);                    // g odd
        const uint32_t x = (f ^ m1) - m1, y = (u ^ m1) - m1, z = (v ^ m1) - m1;     // conditionally negated f, u, v
        g += x & m2; q += y & m2; r += z & m2;
        m1 &= m2;
        zeta = (int32_t)((uint32_t)zeta ^ m1) - 1;            // -zeta - 2 when swapping, else zeta - 1
        f += g & m1; u += q & m1; v += r & m1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return zeta;
}
// [f, g] <- t * [f, g] / 2^30  (exact)
ZKV_HD void modinv30_update_fg(int32_t (&f)[9], int32_t (&g)[9], const ModInv30Trans& t) {
    const int32_t M30 = (int32_t)(0x3fffffffu);
    int64_t cf = (int64_t)t.u * f[0] + (int64_t)t.v * g[0];
    int64_t cg = (int64_t)t.q * f[0] + (int64_t)t.r * g[0];
    cf >>= 30; cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cf += (int64_t)t.u * f[i] + (int64_t)t.v * g[i];
        cg += (int64_t)t.q * f[i] + (int64_t)t.r * g[i];
        f[i - 1] = (int32_t)cf & M30; cf >>= 30;
        g[i - 1] = (int32_t)cg & M30; cg >>= 30;
    }
    f[8] = (int32_t)cf; g[8] = (int32_t)cg;
}
// [d, e] <- t * [d, e] / 2^30 mod m, both kept in (-2m, m)
ZKV_HD void modinv30_update_de(int32_t (&d)[9], int32_t (&e)[9], const ModInv30Trans& t, const int32_t (&m)[9], uint32_t m_inv30) {
    const int32_t M30 = (int32_t)(0x3fffffffu);
    const int32_t sd = d[8] >> 31, se = e[8] >> 31;
    int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
    int64_t cd = (int64_t)t.u * d[0] + (int64_t)t.v * e[0];
    int64_t ce = (int64_t)t.q * d[0] + (int64_t)t.r * e[0];
    // multiples of the modulus that clear the low 30 bits
    md -= (int32_t)((m_inv30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
    me -= (int32_t)((m_inv30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    cd += (int64_t)m[0] * md; ce += (int64_t)m[0] * me;
    cd >>= 30; ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cd += (int64_t)t.u * d[i] + (int64_t)t.v * e[i] + (int64_t)m[i] * md;
        ce += (int64_t)t.q * d[i] + (int64_t)t.r * e[i] + (int64_t)m[i] * me;
        d[i - 1] = (int32_t)cd & M30; cd >>= 30;
        e[i - 1] = (int32_t)ce & M30; ce >>= 30;
    }
    d[8] = (int32_t)cd; e[8] = (int32_t)ce;
}
// r in (-2m, m) -> sign * r mod m in [0, m)   (sign < 0: negate)
ZKV_HD void modinv30_normalize(int32_t (&r)[9], int32_t sign, const int32_t (&m)[9]) {
    const int32_t M30 = (int32_t)(0x3fffffffu);
    int32_t add = r[8] >> 31;
    const int32_t neg = sign >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) r[i] = ((r[i] + (m[i] & add)) ^ neg) - neg;
#pragma unroll
    for (int i = 0; i < 8; i++) { r[i + 1] += r[i] >> 30; r[i] &= M30; }
    add = r[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) r[i] += m[i] & add;
#pragma unroll
    for (int i = 0; i < 8; i++) { r[i + 1] += r[i] >> 30; r[i] &= M30; }
}
// x (8 little-endian 32-bit words, any value below 2^256) <- x^-1 mod m in [0, m); 0 for x = 0 mod m.
// m: the odd modulus as nine 30-bit limbs, m_inv30 = m^-1 mod 2^30.
ZKV_HD void modinv30(uint32_t (&x)[8], const int32_t (&m)[9], uint32_t m_inv30) {
    int32_t d[9], e[9], f[9], g[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { d[i] = 0; e[i] = 0; f[i] = m[i]; }
    e[0] = 1;
    // 8 x 32 -> 9 x 30 bits
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 30 * i, w = bit >> 5, s = bit & 31;
        uint32_t lo = x[w] >> s;
        if (s > 2 && w + 1 < 8) lo |= x[w + 1] << (32 - s);
        g[i] = (int32_t)(lo & 0x3fffffffu);
    }
    int32_t zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 20; it++) {
        ModInv30Trans t;
        zeta = modinv30_divsteps(zeta, (uint32_t)f[0], (uint32_t)g[0], t);
        modinv30_update_de(d, e, t, m, m_inv30);
        modinv30_update_fg(f, g, t);
    }
    // g = 0 now and f = +-gcd(x, m): +-1, or +-m for x = 0 mod m, whose "inverse" is defined as 0 (as a^(m-2) gives)
    modinv30_normalize(d, f[8], m);
    int32_t hi = 0, lo = 0x3fffffff;
#pragma unroll
    for (int i = 1; i < 8; i++) { hi |= f[i]; lo &= f[i]; }
    const bool unit = (f[0] == 1 && (hi | f[8]) == 0) || (f[0] == 0x3fffffff && lo == 0x3fffffff && f[8] == -1);
    const int32_t keep = unit ? -1 : 0;
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] &= keep;
    // 9 x 30 -> 8 x 32 bits
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int bit = 32 * i, l = bit / 30, s = bit % 30;
        x[i] = ((uint32_t)d[l] >> s) | ((uint32_t)d[l + 1] << (30 - s));
    }
}

}  // namespace zkv
