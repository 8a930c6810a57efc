// BN254 G1 / G2 group arithmetic, line functions and the final exponentiation for gfx950 kernels.
//
// Semantics follow the EVM precompiles the reference calls (EIP-196/197; call sites
// /root/reference/contracts/src/common/groth16.rs:54-55 ecMul/ecAdd, :121-125 ecPairing):
// G1 y^2 = x^3 + 3, (0,0) = infinity; G2 on the D-type twist y^2 = x^3 + 3/(9+u) and in the order-r subgroup.
#pragma once
#include "zkv_field.h"

namespace zkv {

// ---------------------------------------------------------------- G1 (Jacobian; Z = 0 is infinity)
struct G1J { Fp x, y, z; };
struct G1A { Fp x, y; };

ZKV_HD bool g1_on_curve(const Fp& x, const Fp& y) {
    Fp three = ZKV_FP_THREE;
    return fp_eq(fp_sqr(y), fp_add(fp_mul(fp_sqr(x), x), three));
}
ZKV_HD G1J g1j_infinity() { G1J r; r.x = fp_one(); r.y = fp_one(); r.z = fp_zero(); return r; }
ZKV_HD G1J g1j_dbl(const G1J& p) {
    if (fp_is_zero(p.z)) return p;
    Fp A = fp_sqr(p.x), B = fp_sqr(p.y), C = fp_sqr(B);
    Fp t = fp_sqr(fp_add(p.x, B));
    Fp D = fp_dbl(fp_sub(fp_sub(t, A), C));
    Fp E = fp_add(fp_dbl(A), A), F = fp_sqr(E);
    G1J r;
    r.x = fp_sub(F, fp_dbl(D));
    r.y = fp_sub(fp_mul(E, fp_sub(D, r.x)), fp_dbl(fp_dbl(fp_dbl(C))));
    r.z = fp_dbl(fp_mul(p.y, p.z));
    return r;
}
// p + (qx,qy) affine (q is never infinity); complete: handles p = inf, p = q, p = -q.
ZKV_HD G1J g1j_add_affine(const G1J& p, const Fp& qx, const Fp& qy) {
    if (fp_is_zero(p.z)) { G1J r; r.x = qx; r.y = qy; r.z = fp_one(); return r; }
    Fp z1z1 = fp_sqr(p.z);
    Fp u2 = fp_mul(qx, z1z1), s2 = fp_mul(fp_mul(qy, p.z), z1z1);
    Fp h = fp_sub(u2, p.x), rr = fp_sub(s2, p.y);
    if (fp_is_zero(h)) {
        if (fp_is_zero(rr)) return g1j_dbl(p);
        return g1j_infinity();
    }
    Fp hh = fp_sqr(h), hhh = fp_mul(hh, h), v = fp_mul(p.x, hh);
    G1J r;
    r.x = fp_sub(fp_sub(fp_sqr(rr), hhh), fp_dbl(v));
    r.y = fp_sub(fp_mul(rr, fp_sub(v, r.x)), fp_mul(p.y, hhh));
    r.z = fp_mul(p.z, h);
    return r;
}

ZKV_HD G1J g1j_add(const G1J& p, const G1J& q) {             // complete Jacobian + Jacobian
    if (fp_is_zero(p.z)) return q;
    if (fp_is_zero(q.z)) return p;
    Fp z1z1 = fp_sqr(p.z), z2z2 = fp_sqr(q.z);
    Fp u1 = fp_mul(p.x, z2z2), u2 = fp_mul(q.x, z1z1);
    Fp s1 = fp_mul(fp_mul(p.y, q.z), z2z2), s2 = fp_mul(fp_mul(q.y, p.z), z1z1);
    Fp h = fp_sub(u2, u1), rr = fp_sub(s2, s1);
    if (fp_is_zero(h)) {
        if (fp_is_zero(rr)) return g1j_dbl(p);
        return g1j_infinity();
    }
    Fp hh = fp_sqr(h), hhh = fp_mul(hh, h), v = fp_mul(u1, hh);
    G1J r;
    r.x = fp_sub(fp_sub(fp_sqr(rr), hhh), fp_dbl(v));
    r.y = fp_sub(fp_mul(rr, fp_sub(v, r.x)), fp_mul(s1, hhh));
    r.z = fp_mul(fp_mul(p.z, q.z), h);
    return r;
}

// ---------------------------------------------------------------- G2 (Jacobian over Fp2)
struct G2J { Fp2 x, y, z; };
struct G2A { Fp2 x, y; };

ZKV_HD bool g2_on_twist(const Fp2& x, const Fp2& y) {
    const Fp2C b = ZKV_TWIST_B;
    return f2_eq(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), f2_const(b)));
}
ZKV_HD G2J g2j_infinity() { G2J r; r.x = f2_one(); r.y = f2_one(); r.z = f2_zero(); return r; }
ZKV_HD G2J g2j_dbl(const G2J& p) {
    if (f2_is_zero(p.z)) return p;
    Fp2 A = f2_sqr(p.x), B = f2_sqr(p.y), C = f2_sqr(B);
    Fp2 t = f2_sqr(f2_add(p.x, B));
    Fp2 D = f2_dbl(f2_sub(f2_sub(t, A), C));
    Fp2 E = f2_add(f2_dbl(A), A), F = f2_sqr(E);
    G2J r;
    r.x = f2_sub(F, f2_dbl(D));
    r.y = f2_sub(f2_mul(E, f2_sub(D, r.x)), f2_dbl(f2_dbl(f2_dbl(C))));
    r.z = f2_dbl(f2_mul(p.y, p.z));
    return r;
}
ZKV_HD G2J g2j_add(const G2J& p, const G2J& q) {
    if (f2_is_zero(p.z)) return q;
    if (f2_is_zero(q.z)) return p;
    Fp2 z1z1 = f2_sqr(p.z), z2z2 = f2_sqr(q.z);
    Fp2 u1 = f2_mul(p.x, z2z2), u2 = f2_mul(q.x, z1z1);
    Fp2 s1 = f2_mul(f2_mul(p.y, q.z), z2z2), s2 = f2_mul(f2_mul(q.y, p.z), z1z1);
    Fp2 h = f2_sub(u2, u1), rr = f2_sub(s2, s1);
    if (f2_is_zero(h)) {
        if (f2_is_zero(rr)) return g2j_dbl(p);
        return g2j_infinity();
    }
    Fp2 hh = f2_sqr(h), hhh = f2_mul(hh, h), v = f2_mul(u1, hh);
    G2J r;
    r.x = f2_sub(f2_sub(f2_sqr(rr), hhh), f2_dbl(v));
    r.y = f2_sub(f2_mul(rr, f2_sub(v, r.x)), f2_mul(s1, hhh));
    r.z = f2_mul(f2_mul(p.z, q.z), h);
    return r;
}
ZKV_HD G2J g2j_add_affine(const G2J& p, const Fp2& qx, const Fp2& qy) {
    if (f2_is_zero(p.z)) { G2J r; r.x = qx; r.y = qy; r.z = f2_one(); return r; }
    Fp2 z1z1 = f2_sqr(p.z);
    Fp2 u2 = f2_mul(qx, z1z1), s2 = f2_mul(f2_mul(qy, p.z), z1z1);
    Fp2 h = f2_sub(u2, p.x), rr = f2_sub(s2, p.y);
    if (f2_is_zero(h)) {
        if (f2_is_zero(rr)) return g2j_dbl(p);
        return g2j_infinity();
    }
    Fp2 hh = f2_sqr(h), hhh = f2_mul(hh, h), v = f2_mul(p.x, hh);
    G2J r;
    r.x = f2_sub(f2_sub(f2_sqr(rr), hhh), f2_dbl(v));
    r.y = f2_sub(f2_mul(rr, f2_sub(v, r.x)), f2_mul(p.y, hhh));
    r.z = f2_mul(p.z, h);
    return r;
}
// psi = twist o Frobenius o untwist on Jacobian coordinates
ZKV_HD G2J g2j_psi(const G2J& p) {
    const Fp2C G[6] = ZKV_FROB1;
    G2J r; r.x = f2_mul(f2_conj(p.x), f2_const(G[2])); r.y = f2_mul(f2_conj(p.y), f2_const(G[3])); r.z = f2_conj(p.z);
    return r;
}
ZKV_HD bool g2j_eq(const G2J& a, const G2J& b) {
    bool ia = f2_is_zero(a.z), ib = f2_is_zero(b.z);
    if (ia || ib) return ia && ib;
    Fp2 za2 = f2_sqr(a.z), zb2 = f2_sqr(b.z);
    if (!f2_eq(f2_mul(a.x, zb2), f2_mul(b.x, za2))) return false;
    return f2_eq(f2_mul(a.y, f2_mul(zb2, b.z)), f2_mul(b.y, f2_mul(za2, a.z)));
}
// [u]P for the BN parameter u (63 bits), P affine, complete formulas (P may have small order).  Signed digits (NAF of u,
// 24 non-zero digits instead of 28 set bits): a -1 digit adds -P = (x, -y).
ZKV_HD G2J g2_mul_u(const Fp2& px, const Fp2& py) {
    const int8_t NAF[ZKV_U_NAF_LEN] = ZKV_U_NAF;
    const Fp2 ny = f2_neg(py);
    G2J acc; acc.x = px; acc.y = py; acc.z = f2_one();
#pragma unroll 1
    for (int i = ZKV_U_NAF_LEN - 2; i >= 0; i--) {
        acc = g2j_dbl(acc);
        const int d = NAF[i];
        if (d != 0) acc = g2j_add_affine(acc, px, d > 0 ? py : ny);
    }
    return acc;
}
// Order-r subgroup test of an on-twist point, equivalent to EIP-197's [r]Q = O:
//   [u+1]Q + psi([u]Q) + psi^2([u]Q) == psi^3([2u]Q)
// (checked against the literal [r]Q = O on in-subgroup, random, small-order and mixed points in tests).
ZKV_HD bool g2_in_subgroup(const Fp2& qx, const Fp2& qy) {
    G2J a = g2_mul_u(qx, qy);
    G2J b = g2j_psi(a);
    G2J c = g2j_psi(b);
    G2J lhs = g2j_add(g2j_add(g2j_add_affine(a, qx, qy), b), c);
    G2J rhs = g2j_psi(g2j_psi(g2j_psi(g2j_dbl(a))));
    return g2j_eq(lhs, rhs);
}

// ---------------------------------------------------------------- Miller-loop line functions
// T in homogeneous projective coordinates; line = l0 * yP + l1 * xP * w + l3 * w^3 (up to an Fp2 factor,
// which the final exponentiation removes).
struct G2H { Fp2 x, y, z; };

ZKV_HD void line_dbl(G2H& T, Fp2& l0, Fp2& l1, Fp2& l3) {
    const Fp2C b3c = ZKV_TWIST_3B;
    const Fp2 b3 = f2_const(b3c);
    Fp2 a = f2_half(f2_mul(T.x, T.y));
    Fp2 b = f2_sqr(T.y), c = f2_sqr(T.z);
    Fp2 e = f2_mul(b3, c);                              // 3 b' Z^2
    Fp2 f = f2_add(f2_dbl(e), e);                       // 9 b' Z^2
    Fp2 g = f2_half(f2_add(b, f));
    Fp2 h = f2_sub(f2_sqr(f2_add(T.y, T.z)), f2_add(b, c));   // 2YZ
    Fp2 j = f2_sqr(T.x);
    Fp2 e2 = f2_sqr(e);
    l0 = f2_neg(h); l1 = f2_add(f2_dbl(j), j); l3 = f2_sub(e, b);
    T.x = f2_mul(a, f2_sub(b, f));
    T.y = f2_sub(f2_sqr(g), f2_add(f2_dbl(e2), e2));
    T.z = f2_mul(b, h);
}
ZKV_HD void line_add(G2H& T, const Fp2& qx, const Fp2& qy, Fp2& l0, Fp2& l1, Fp2& l3) {
    Fp2 theta = f2_sub(T.y, f2_mul(qy, T.z));
    Fp2 lambda = f2_sub(T.x, f2_mul(qx, T.z));
    Fp2 c = f2_sqr(theta), d = f2_sqr(lambda), e = f2_mul(lambda, d);
    Fp2 f = f2_mul(T.z, c), g = f2_mul(T.x, d);
    Fp2 h = f2_sub(f2_add(e, f), f2_dbl(g));
    l0 = lambda; l1 = f2_neg(theta); l3 = f2_sub(f2_mul(theta, qx), f2_mul(lambda, qy));
    T.x = f2_mul(lambda, h);
    T.y = f2_sub(f2_mul(theta, f2_sub(g, h)), f2_mul(e, T.y));
    T.z = f2_mul(T.z, e);
}
ZKV_HD void g2_frob_affine(Fp2& x, Fp2& y, const Fp2& qx, const Fp2& qy) {
    const Fp2C G[6] = ZKV_FROB1;
    x = f2_mul(f2_conj(qx), f2_const(G[2])); y = f2_mul(f2_conj(qy), f2_const(G[3]));
}
// pi^2 on the twist: multiplication by Fp constants
ZKV_HD void g2_frob2_affine(Fp2& x, Fp2& y, const Fp2& qx, const Fp2& qy) {
    const Fp G[6] = ZKV_FROB2;
    x = f2_mul_fp(qx, G[2]); y = f2_mul_fp(qy, G[3]);
}

// Affine stepping used only when building the fixed-Q line tables at context set-up (gamma, delta):
// slope form  line/yP = 1 + (nl * xP/yP) w + (c * 1/yP) w^3  with nl = -lambda, c = lambda xT - yT.
struct LineAff { Fp2 nl, c; };
struct LineAffC { Fp2C nl, c; };       // table layout (both components), read by every kernel variant
ZKV_HD LineAff aff_dbl(G2A& T) {
    Fp2 x2 = f2_sqr(T.x);
    Fp2 lam = f2_mul(f2_add(f2_dbl(x2), x2), f2_inv(f2_dbl(T.y)));
    LineAff l; l.nl = f2_neg(lam); l.c = f2_sub(f2_mul(lam, T.x), T.y);
    Fp2 x3 = f2_sub(f2_sqr(lam), f2_dbl(T.x));
    Fp2 y3 = f2_sub(f2_mul(lam, f2_sub(T.x, x3)), T.y);
    T.x = x3; T.y = y3;
    return l;
}
ZKV_HD LineAff aff_add(G2A& T, const Fp2& qx, const Fp2& qy) {
    Fp2 lam = f2_mul(f2_sub(qy, T.y), f2_inv(f2_sub(qx, T.x)));
    LineAff l; l.nl = f2_neg(lam); l.c = f2_sub(f2_mul(lam, T.x), T.y);
    Fp2 x3 = f2_sub(f2_sub(f2_sqr(lam), T.x), qx);
    Fp2 y3 = f2_sub(f2_mul(lam, f2_sub(T.x, x3)), T.y);
    T.x = x3; T.y = y3;
    return l;
}

}  // namespace zkv
