// Per-proof stages of the batched Groth16 verify path, one proof per lane.
//
// Mirrors, stage by stage, the reference control flow (paths under /root/reference/contracts/src):
//   stage PREP    risc0/verifier.rs:146-179 / sp1/verifier.rs:58-94  selector + strict length + seal words,
//                 risc0/types.rs:44-94 claim digest, risc0/crypto.rs:95-110 split_digest,
//                 sp1/types.rs:22-38 public-input hashing; common/groth16.rs:32 signal range check;
//                 groth16.rs:75-84 negate_g1; EIP-196/197 coordinate / on-curve validation
//   stage MSM     common/groth16.rs:51-58 compute_vk_x (ecMul/ecAdd loop) -> fixed-base windowed sum
//   stage G2CHK   EIP-197 subgroup validation of B inside the ecPairing call (groth16.rs:121-125)
//   stage MILLER  + FINALEXP   the ecPairing precompile itself (groth16.rs:109-128)
// The functions are __host__ __device__ so tests/host_sim can run the identical code on the CPU.
#pragma once
#include "zkv_tower_mem.h"
#include "zkv_sha256.h"

namespace zkv {

enum : uint8_t {
    ST_OK = 0, ST_VERIFICATION_FAILED = 1, ST_INVALID_INITIALIZATION = 2, ST_ALREADY_INITIALIZED = 3,
    ST_INVALID_PROOF_DATA = 4, ST_SELECTOR_MISMATCH = 5,
    ST_BAD_CALLDATA = 6            // wire layer only: calldata the contract's router cannot decode
};
enum : uint32_t { FL_ALIVE = 1u, FL_A_INF = 2u, FL_B_INF = 4u, FL_C_INF = 8u, FL_L_INF = 16u };

constexpr int N_LINES = 88;          // 65 doublings + 21 NAF additions + 2 Frobenius additions
constexpr int MSM_MAX_WINDOWS = 32;  // 8-bit windows over a 256-bit scalar
constexpr int MSM_DIGITS = 256;
constexpr int MAX_IC = 6;
constexpr int MAX_VAR = 5;           // per-proof signals (risc0 and sp1 use 2; a generic key uses all n_ic - 1)

// Raw verification key handed to the set-up kernel: canonical values as 8 x 32-bit little-endian limbs.
struct VkRaw {
    uint32_t alpha[2][8];
    uint32_t beta[4][8], gamma[4][8], delta[4][8];   // x_re, x_im, y_re, y_im
    uint32_t ic[MAX_IC][2][8];
    uint32_t n_ic;
    uint32_t fixed_scalar[MAX_IC][8];   // for IC index i >= 1: per-context scalar (risc0 control root / id)
    uint32_t is_fixed[MAX_IC];
    uint32_t n_var;                     // number of per-proof scalars
    uint32_t var_ic[MAX_VAR];           // their IC indices
    uint32_t var_windows[MAX_VAR];      // 8-bit windows per per-proof scalar (16 for 128-bit, 32 for 256-bit)
};

// Device-resident tables derived from the VK at context set-up.
struct VkTables {
    G1A base; uint32_t base_inf; uint32_t n_var; uint32_t var_windows[MAX_VAR];
    uint32_t vk_valid;                   // every VK point is a valid precompile input (else every proof is rejected)
    uint32_t skip_fixed[2];              // gamma / delta is the point at infinity: that pair contributes 1
    G1A msm[MAX_VAR][MSM_MAX_WINDOWS][MSM_DIGITS];   // msm[b][w][d] = d * 256^w * IC_var[b]   (d = 0 unused); 2.6 MB, L2-resident rows
    LineAffC lines[2][N_LINES];          // gamma, delta: slope-form lines of the fixed-Q Miller loop
    uint32_t f_alpha_beta[96];           // Miller value of (alpha, beta): Fp12 as g0 g1 g2 h0 h1 h2, (c0, c1) each
};

// Wide windows of the vk_x stage for big batches: entry (row0[b] + w, d) = d * 65536^w * IC_var[b], d = 1 .. 65535 -- half the additions of
// the 8-bit walk.  4 MB per row (SP1: 32 rows, RISC Zero: 16), its own allocation built at context set-up from the 8-bit rows
// (setup_msm16_chunk); tab == nullptr: not built (ZKV_MSM_WINDOW_BITS=8, or a generic key with more than MSM16_MAX_ROWS rows).
constexpr uint32_t MSM16_MAX_ROWS = 32;
struct Msm16 { const G1A* tab; uint32_t row0[MAX_VAR]; };

// Verifier sets: raw parameters, set-up constants and the per-instance device table (see setup_instance).
struct InstRaw { uint8_t control_root[32]; uint8_t control_id[32]; };
struct InstConsts { uint8_t tag[32]; uint8_t vk_digest[32]; };   // sha256("risc0.Groth16ReceiptVerifierParameters"), VK digest
struct InstTab { G1A base; uint32_t base_inf, selector_be, fail, pad; };

struct PrepOut {
    Fp ax, ay, cx, cy; Fp2 bx, by;
    uint32_t s[MAX_VAR][8];
    uint32_t flags;
};
struct G1Norm { Fp axs, ays, lxs, lys, cxs, cys; };   // x/y and 1/y of A', L = vk_x, C

ZKV_HD int8_t ate_naf(int i) {
    const int8_t NAF[ZKV_ATE_NAF_LEN] = ZKV_ATE_NAF;
    return NAF[i];
}

// ---------------------------------------------------------------- raw 256-bit helpers
ZKV_HD void load_be256(uint32_t limbs[8], const uint8_t* p) {
#pragma unroll 1
    for (int i = 0; i < 8; i++) limbs[7 - i] = load_be32(p + 4 * i);
}
ZKV_HD bool raw_is_zero(const uint32_t* a) {
    uint32_t o = 0;
    for (int i = 0; i < 8; i++) o |= a[i];
    return o == 0;
}
ZKV_HD bool raw_lt_p(const uint32_t* a) { const uint32_t P[8] = ZKV_FP_P_LIMBS; return !u256_geq(a, P); }
ZKV_HD bool raw_lt_r(const uint32_t* a) { const uint32_t R[8] = ZKV_FR_R_LIMBS; return !u256_geq(a, R); }

struct Risc0Consts { uint32_t tag_output[8]; uint32_t claim_mid[8]; uint32_t post[8]; };   // see risc0_claim_digest

#if !defined(ZKV_PAIRED)
// ---------------------------------------------------------------- stage PREP (shared part)
// words: the 8 proof words a.x a.y b.x_im b.x_re b.y_im b.y_re c.x c.y as raw limbs.
// Returns false => VerificationFailed (a precompile would reject the point encoding).
ZKV_HD bool prep_points(uint32_t w[8][8], bool negate_a, PrepOut& o) {
    const uint32_t P[8] = ZKV_FP_P_LIMBS;
    if (negate_a && !(raw_is_zero(w[0]) && raw_is_zero(w[1]))) {      // groth16.rs:75-84, Q.wrapping_sub(y)
        uint32_t br = 0;
        for (int i = 0; i < 8; i++) w[1][i] = subb(P[i], w[1][i], br);
    }
    bool ok = true;
    for (int k = 0; k < 8; k++) ok = ok && raw_lt_p(w[k]);
    if (!ok) return false;
    uint32_t fl = 0;
    if (raw_is_zero(w[0]) && raw_is_zero(w[1])) {
        fl |= FL_A_INF; o.ax = fp_zero(); o.ay = fp_zero();
    } else {
        o.ax = fp_from_raw(w[0]); o.ay = fp_from_raw(w[1]);
        if (!g1_on_curve(o.ax, o.ay)) return false;
    }
    if (raw_is_zero(w[6]) && raw_is_zero(w[7])) {
        fl |= FL_C_INF; o.cx = fp_zero(); o.cy = fp_zero();
    } else {
        o.cx = fp_from_raw(w[6]); o.cy = fp_from_raw(w[7]);
        if (!g1_on_curve(o.cx, o.cy)) return false;
    }
    if (raw_is_zero(w[2]) && raw_is_zero(w[3]) && raw_is_zero(w[4]) && raw_is_zero(w[5])) {
        fl |= FL_B_INF; o.bx = f2_zero(); o.by = f2_zero();
    } else {
        o.bx.c1 = fp_from_raw(w[2]); o.bx.c0 = fp_from_raw(w[3]);    // wire order (im, re), SURVEY a8
        o.by.c1 = fp_from_raw(w[4]); o.by.c0 = fp_from_raw(w[5]);
        if (!g2_on_twist(o.bx, o.by)) return false;
    }
    o.flags = fl | FL_ALIVE;
    return true;
}

// risc0/types.rs:44-94: claim digest from (image_id, journal_digest).  tag_output = sha256("risc0.Output"),
// claim_mid = SHA-256 state after the constant first block of the claim message
// (sha256("risc0.ReceiptClaim") || input = 0^32), post = SYSTEM_STATE_ZERO_DIGEST words.

ZKV_HD void risc0_claim_digest(const Risc0Consts& k, const uint8_t* image_id, const uint8_t* journal, uint32_t h[8]) {
    uint32_t w[16], out[8];
    // Output::digest: tag || journal || 0^32 || 0x0200   (98 bytes)
    sha256_init(out);
    for (int i = 0; i < 8; i++) { w[i] = k.tag_output[i]; w[8 + i] = load_be32(journal + 4 * i); }
    sha256_compress(out, w);
    for (int i = 0; i < 8; i++) w[i] = 0;
    w[8] = 0x02008000u;                                   // 02 00 | 0x80 pad
    for (int i = 9; i < 15; i++) w[i] = 0;
    w[15] = 98u * 8u;
    sha256_compress(out, w);
    // ReceiptClaim::digest: tag || input || pre || post || output || 0^8 || 0x0400   (170 bytes)
    for (int i = 0; i < 8; i++) h[i] = k.claim_mid[i];
    for (int i = 0; i < 8; i++) { w[i] = load_be32(image_id + 4 * i); w[8 + i] = k.post[i]; }
    sha256_compress(h, w);
    for (int i = 0; i < 8; i++) w[i] = out[i];
    w[8] = 0; w[9] = 0;
    w[10] = 0x04008000u;
    for (int i = 11; i < 15; i++) w[i] = 0;
    w[15] = 170u * 8u;
    sha256_compress(h, w);
}
// risc0/crypto.rs:95-110: low/high 128-bit halves of the byte-reversed digest as scalars
ZKV_HD void risc0_split_digest(const uint32_t h[8], uint32_t lo[8], uint32_t hi[8]) {
    for (int i = 0; i < 4; i++) {
        lo[i] = __builtin_bswap32(h[i]); hi[i] = __builtin_bswap32(h[4 + i]);
        lo[4 + i] = 0; hi[4 + i] = 0;
    }
}

// ---------------------------------------------------------------- stage MSM + G1 normalisation
// vk_x = base + sum_b s_b * IC_var[b] (groth16.rs:51-58), then x/y and 1/y of A', L, C with one inversion.
// `base` = IC[0] + the per-verifier-instance signals: the context's own (vk.base) or, for a verifier set, the instance's.
// word(b, k): limb k of the per-proof scalar b.  The kernels read it from the proof's workspace row where it is needed (a private copy of
// the scalars, indexed by the run-time b, was k_msm's 432-byte scratch frame); the host builds and k_vk_x pass their PrepOut.
template <class WORD> ZKV_HD G1J msm_accumulate_w(const VkTables& vk, WORD word, const G1A& base, uint32_t base_inf) {
    G1J acc;
    if (base_inf) acc = g1j_infinity();
    else { acc.x = base.x; acc.y = base.y; acc.z = fp_one(); }
#pragma unroll 1
    for (uint32_t b = 0; b < vk.n_var; b++) {
#pragma unroll 1
        for (uint32_t w = 0; w < vk.var_windows[b]; w++) {
            uint32_t d = (word(b, w >> 2) >> ((w & 3) * 8)) & 255u;
            if (d) {
                const G1A& e = vk.msm[b][w][d];
                acc = g1j_add_affine(acc, e.x, e.y);
            }
        }
    }
    return acc;
}
// The same walk over 16-bit windows (Msm16): one 64-byte gather and one addition per 16 bits of a scalar.
template <class WORD> ZKV_HD G1J msm_accumulate_w16(const VkTables& vk, const Msm16& m, WORD word, const G1A& base, uint32_t base_inf) {
    G1J acc;
    if (base_inf) acc = g1j_infinity();
    else { acc.x = base.x; acc.y = base.y; acc.z = fp_one(); }
#pragma unroll 1
    for (uint32_t b = 0; b < vk.n_var; b++) {
        const uint32_t nw = vk.var_windows[b];                  // 8-bit windows of this scalar (0: IC_b is infinity)
#pragma unroll 1
        for (uint32_t w = 0; 2 * w < nw; w++) {
            uint32_t d = (word(b, w >> 1) >> ((w & 1) * 16)) & 0xffffu;
            if (2 * w + 1 >= nw) d &= 0xffu;                    // an odd number of 8-bit windows: the last row has no upper half
            if (d) {
                const G1A e = m.tab[((size_t)(m.row0[b] + w) << 16) + d];
                acc = g1j_add_affine(acc, e.x, e.y);
            }
        }
    }
    return acc;
}
ZKV_HD G1J msm_accumulate(const VkTables& vk, const PrepOut& in, const G1A& base, uint32_t base_inf) {
    return msm_accumulate_w(vk, [&](uint32_t b, uint32_t k) { return in.s[b][k]; }, base, base_inf);
}
ZKV_HD G1J msm_accumulate(const VkTables& vk, const PrepOut& in) { return msm_accumulate(vk, in, vk.base, vk.base_inf); }
ZKV_HD void msm_normalize_acc(const G1J& acc, const PrepOut& in, uint32_t& flags, G1Norm& out);
ZKV_HD void msm_normalize(const VkTables& vk, const PrepOut& in, uint32_t& flags, G1Norm& out, const G1A& base, uint32_t base_inf) {
    msm_normalize_acc(msm_accumulate(vk, in, base, base_inf), in, flags, out);
}
ZKV_HD void msm_normalize_acc(const G1J& acc, const PrepOut& in, uint32_t& flags, G1Norm& out) {      // uses in.ax ay cx cy only
    Fp one = fp_one();
    bool linf = fp_is_zero(acc.z), ainf = (flags & FL_A_INF) != 0, cinf = (flags & FL_C_INF) != 0;
    if (linf) flags |= FL_L_INF;
    Fp ya = ainf ? one : in.ay, yl = linf ? one : acc.y, yc = cinf ? one : in.cy;
    Fp t = fp_mul(ya, yl);
    Fp inv = fp_inv(fp_mul(t, yc));
    Fp iyc = fp_mul(inv, t);
    Fp u = fp_mul(inv, yc);            // 1/(ya yl)
    Fp iyl = fp_mul(u, ya), iya = fp_mul(u, yl);
    out.axs = fp_mul(in.ax, iya); out.ays = iya;
    out.cxs = fp_mul(in.cx, iyc); out.cys = iyc;
    Fp z2 = fp_sqr(acc.z);
    out.lxs = fp_mul(fp_mul(acc.x, acc.z), iyl);         // (X/Z^2) / (Y/Z^3) = X Z / Y
    out.lys = fp_mul(fp_mul(z2, acc.z), iyl);            // Z^3 / Y
}
ZKV_HD void msm_normalize(const VkTables& vk, const PrepOut& in, uint32_t& flags, G1Norm& out) { msm_normalize(vk, in, flags, out, vk.base, vk.base_inf); }

#endif  // !ZKV_PAIRED (PREP and MSM run one proof per lane)

// ---------------------------------------------------------------- stage MILLER
// Shared-accumulator Miller loop over (A',B) [variable Q, projective], (L,gamma), (C,delta) [fixed Q,
// precomputed slope lines]; multiplied by the precomputed Miller value of (alpha, beta).
// fm: the Fp12 accumulator slot, tm: the 3 x Fp2 running point T (both LDS on the device).
template <class RF> ZKV_HD void fixed_line_mul(RF fm, const LineAffC& L, const Fp& xs, const Fp& ys) {
    Fp2 c3 = f2_mul_fp(f2_const(L.nl), xs), c4 = f2_mul_fp(f2_const(L.c), ys);
    f12m_mul_by_134(fm, &c3, &c4);
}
template <class RF> ZKV_HD void var_line_mul(RF fm, const Fp2& l0, const Fp2& l1, const Fp2& l3, const Fp& xs, const Fp& ys) {
    Fp2 c3 = f2_mul_fp(l1, xs), c4 = f2_mul_fp(l3, ys);
    f12m_mul_by_034(fm, &l0, &c3, &c4);
}
// vkp: the context's tables, or nullptr for a single variable pair without fixed pairs (the ecPairing seam, e(alpha, beta) at set-up).
// Does psi^3(B) = -T hold for the running point T left by the loop, T != O?  <=> B in the order-r subgroup (see miller_loop_p).
template <class RT> ZKV_HD bool miller_point_closes(RT tm, const Fp2& bx, const Fp2& by) {
    const Fp2C G3[6] = ZKV_FROB3;
    const Fp2 tz = m_ld_f2(tm, 2);
    const Fp2 x3 = f2_mul(f2_mul(f2_conj(bx), f2_const(G3[2])), tz);       // psi^3(B) scaled to T's Z
    const Fp2 y3 = f2_mul(f2_mul(f2_conj(by), f2_const(G3[3])), tz);
    return !f2_is_zero(tz) && f2_eq(m_ld_f2(tm, 0), x3) && f2_is_zero(f2_add(m_ld_f2(tm, 1), y3));
}
// check_b: also step the point for A = infinity and return the subgroup verdict for B (miller_loop_p explains why this is one).
template <class RF, class RT>
ZKV_HD bool miller_loop_m(const VkTables* vkp, uint32_t flags, const G1Norm& n, const Fp2& bx, const Fp2& by, RF fm, RT tm, bool check_b = false) {
    const bool with_fixed = vkp != nullptr;
    const VkTables* vkq = with_fixed ? vkp : nullptr;
    const bool do_ab = !(flags & (FL_A_INF | FL_B_INF));
    const bool do_t = do_ab || (check_b && !(flags & FL_B_INF));
    bool do_l = with_fixed && !(flags & FL_L_INF) && !vkq->skip_fixed[0], do_c = with_fixed && !(flags & FL_C_INF) && !vkq->skip_fixed[1];
    const LineAffC* lines0 = with_fixed ? vkq->lines[0] : nullptr;
    const LineAffC* lines1 = with_fixed ? vkq->lines[1] : nullptr;
    f12m_set_one(fm);
    m_st_f2(tm, 0, bx); m_st_f2(tm, 1, by); m_st_f2(tm, 2, f2_one());
    Fp2 nby = f2_neg(by);
    Fp2 l0, l1, l3;
    int li = 0;
#pragma unroll 1
    for (int i = ZKV_ATE_NAF_LEN - 2; i >= 0; i--) {
        if (i != ZKV_ATE_NAF_LEN - 2) f12m_sqr(fm);
        if (do_t) {
            g2m_line_dbl(tm, &l0, &l1, &l3);
            if (do_ab) var_line_mul(fm, l0, l1, l3, n.axs, n.ays);
        }
        if (do_l) fixed_line_mul(fm, lines0[li], n.lxs, n.lys);
        if (do_c) fixed_line_mul(fm, lines1[li], n.cxs, n.cys);
        li++;
        int d = ate_naf(i);
        if (d != 0) {
            if (do_t) {
                Fp2 qy = d > 0 ? by : nby;
                g2m_line_add(tm, &bx, &qy, &l0, &l1, &l3);
                if (do_ab) var_line_mul(fm, l0, l1, l3, n.axs, n.ays);
            }
            if (do_l) fixed_line_mul(fm, lines0[li], n.lxs, n.lys);
            if (do_c) fixed_line_mul(fm, lines1[li], n.cxs, n.cys);
            li++;
        }
    }
    Fp2 qx[2], qy[2];
    g2_frob_affine(qx[0], qy[0], bx, by);
    g2_frob2_affine(qx[1], qy[1], bx, by);
    qy[1] = f2_neg(qy[1]);
#pragma unroll 1
    for (int s = 0; s < 2; s++) {
        if (do_t) {
            g2m_line_add(tm, &qx[s], &qy[s], &l0, &l1, &l3);
            if (do_ab) var_line_mul(fm, l0, l1, l3, n.axs, n.ays);
        }
        if (do_l) fixed_line_mul(fm, lines0[li], n.lxs, n.lys);
        if (do_c) fixed_line_mul(fm, lines1[li], n.cxs, n.cys);
        li++;
    }
    return !(check_b && do_t) || miller_point_closes(tm, bx, by);
}

#if defined(ZKV_PAIRED)
// The same loop for the lane-pair kernel, as ONE flat sequence of 88 line steps in which every Fp12-level routine is inlined at
// exactly one place.  Why: a non-inlined Fp12 routine keeps its cross-call values in callee-saved VGPRs and must save / restore them
// around its body (27 dwords per f12m_sqr call, about 5,000 scratch accesses per proof and lane, every one of which reaches HBM: 3.4 GB
// per 2^16-proof launch, profiles/round1_pair_pmc_traffic.json).  Inlined into the kernel -- which has no caller to preserve anything
// for -- the routines need no frame; only fp_mul / f2_mul_lane stay calls (leaf functions inside the caller-saved registers).
// Per-proof constants (x/y and 1/y of the three G1 points, B) are re-read from their workspace rows where they are used instead of
// being held in registers across the whole loop.
struct SoaRef {                         // word k of this proof's (this lane's) value at base[k * stride] + lane offset
    // base and stride are wave-uniform, the lane's part of the address is one 32-bit byte offset: every load is
    // `global_load_dword v, v_off, s[base]` with the base advanced on the scalar unit.  (With a per-lane 64-bit pointer the compiler
    // hoisted ten loop-invariant per-lane addresses out of the Miller loop and spilled them: 5.5 GB of scratch reloads per launch.)
    const uint32_t* p; size_t stride; uint32_t off;
    ZKV_HD Fp fp(int word0) const {
        Fp r;
#pragma unroll
        for (int k = 0; k < 8; k++) r.v[k] = *(const uint32_t*)((const char*)(p + (size_t)(word0 + k) * stride) + off);
        return r;
    }
};
// norm: axs ays lxs lys cxs cys at words 0 8 16 24 32 40; bsrc: this lane's component of B.x at word 0 and of B.y at word 16.
//
// check_b: the loop doubles as the order-r subgroup test of B (EIP-197), and returns its verdict.  After the 88 steps the running
// point is T = [6u+2]B + psi(B) - psi^2(B), and (6u+2) + p - p^2 + p^3 = 0 mod r, so B in G2 implies T = -psi^3(B) -- the reason the
// optimal-ate pairing has no fourth line.  The converse holds as well: h(psi)B = O for h = (6u+2) + X - X^2 + X^3 implies
// ord(B) | gcd(Res(h, X^2 - tX + p), #E'(Fp2)), and that gcd is r (tools/check_g2_vector.py; the classical test vector
// (u+1, u, u, -2u) of g2_in_subgroup passes the same computation).  A point outside G2 may drive the incomplete tangent / chord formulas
// into an exceptional case (T = O, T = +-B, a 2-torsion T); every one of them leaves Z = 0, and Z = 0 is absorbing for both formulas
// (Z3 = 2 Y^3 Z, Z3 = Z lambda^3), so such a point ends with Z = 0 and is rejected, while a point of G2 (prime order r, larger than
// every partial scalar) never meets one and ends with Z != 0.  With check_b the point is stepped for A = infinity too (the precompile
// validates B whether or not the pair contributes); only the line products are skipped then.
template <class RF, class RT>
ZKV_HD bool miller_loop_p(const VkTables* vkp, uint32_t flags, SoaRef norm, SoaRef bsrc, RF fm, RT tm, bool check_b = false) {
    const uint8_t KIND[ZKV_MILLER_STEPS] = ZKV_MILLER_STEP_KIND;
    const bool with_fixed = vkp != nullptr;
    const bool do_ab = !(flags & (FL_A_INF | FL_B_INF));
    const bool do_t = do_ab || (check_b && !(flags & FL_B_INF));
    const bool do_l = with_fixed && !(flags & FL_L_INF) && !vkp->skip_fixed[0], do_c = with_fixed && !(flags & FL_C_INF) && !vkp->skip_fixed[1];
    f12m_set_one(fm);
    { Fp2 bx, by; bx.h = bsrc.fp(0); by.h = bsrc.fp(16); m_st_f2(tm, 0, bx); m_st_f2(tm, 1, by); m_st_f2(tm, 2, f2_one()); }
    const uint32_t slot = zkv_wave_slot_parity();
#pragma unroll 1
    for (int li = 0; li < ZKV_MILLER_STEPS; li++) {
        const int kind = KIND[li];
        zkv_fair_share(slot);                  // the two wavefronts of a SIMD take turns at priority (zkv_field.h)
#if defined(__HIP_DEVICE_COMPILE__)
        // the lane offsets are re-read through an opaque move every step: otherwise the 64 per-lane addresses of the step's loads are
        // hoisted out of the loop as invariants, and ten of them end up in a scratch frame (5.5 GB of reloads per 2^20-proof launch)
        asm volatile("" : "+v"(norm.off), "+v"(bsrc.off));
#endif
        if (kind == 0 && li != 0) { ZKV_MARK("begin sqr"); f12m_sqr_body(fm); ZKV_MARK("end sqr"); }
        if (do_t) {
            Fp2 l0, l1, l3;
            G2H T; T.x = m_ld_f2(tm, 0); T.y = m_ld_f2(tm, 1); T.z = m_ld_f2(tm, 2);
            if (kind == 0) { ZKV_MARK("begin linedbl"); line_dbl(T, l0, l1, l3); ZKV_MARK("end linedbl"); }
            else {
                Fp2 qx, qy; qx.h = bsrc.fp(0); qy.h = bsrc.fp(16);
                if (kind == 2) qy = f2_neg(qy);
                else if (kind == 3) { Fp2 x, y; g2_frob_affine(x, y, qx, qy); qx = x; qy = y; }
                else if (kind == 4) { Fp2 x, y; g2_frob2_affine(x, y, qx, qy); qx = x; qy = f2_neg(y); }
                ZKV_MARK("begin lineadd"); line_add(T, qx, qy, l0, l1, l3); ZKV_MARK("end lineadd");
            }
            m_st_f2(tm, 0, T.x); m_st_f2(tm, 1, T.y); m_st_f2(tm, 2, T.z);
            if (do_ab) {
                ZKV_MARK("begin mul034");
                const Fp2 c3 = f2_mul_fp(l1, norm.fp(0)), c4 = f2_mul_fp(l3, norm.fp(8));
                f12m_mul_by_034_body(fm, l0, c3, c4);
                ZKV_MARK("end mul034");
            }
        }
#pragma unroll 1
        for (int j = 0; j < 2; j++) {
            if (j == 0 ? !do_l : !do_c) continue;
            // the step's line is the same for every lane: 8 vector loads of one broadcast address.  Reading it through the constant
            // address space instead (two s_load_dwordx16 into SGPRs; commit 'Line tables: scalar-load variant built and measured') was
            // slower -- 125.3 against 123.8 ms per 2^20 proofs: 32 more live SGPRs and a scalar wait in front of every line product.
            ZKV_MARK("begin mul134");
            const LineAffC& L = vkp->lines[j][li];
            const Fp2 c3 = f2_mul_fp(f2_const(L.nl), norm.fp(16 + 16 * j)), c4 = f2_mul_fp(f2_const(L.c), norm.fp(24 + 16 * j));
            f12m_mul_by_134_body(fm, c3, c4);
            ZKV_MARK("end mul134");
        }
    }
    if (!check_b || !do_t) return true;
    Fp2 qx, qy; qx.h = bsrc.fp(0); qy.h = bsrc.fp(16);
    return miller_point_closes(tm, qx, qy);
}
// The Miller loop of the aggregate check (zkv_agg.h) for G proofs of one lane pair that share the accumulator: f <- f^2 once per
// doubling step, then each proof's line -- only the variable pairs (r A'_p, B_p); the fixed pairs are taken once per sub-batch.  The
// running points live in HBM rows (tq; proof p's rows p * `step` bytes after proof 0's, as in norm and bsrc) and pass through registers
// only while their line is formed: LDS holds nothing but f, which keeps two wavefronts per SIMD.  mask bit p: proof p takes part
// (alive, B finite); abmask bit p: its pair contributes (A finite as well).  Returns bit p set when proof p's B passed the
// subgroup test the loop doubles as (miller_loop_p); a proof outside the mask reports set.
template <int G, class RF>
ZKV_HD uint32_t miller_loop_pg(uint32_t mask, uint32_t abmask, SoaRef norm, SoaRef bsrc, SoaRW tq, uint32_t step, RF fm) {
    const uint8_t KIND[ZKV_MILLER_STEPS] = ZKV_MILLER_STEP_KIND;
    f12m_set_one(fm);
#pragma unroll 1
    for (uint32_t p = 0; p < (uint32_t)G; p++) {
        if (!((mask >> p) & 1u)) continue;
        SoaRef b = bsrc; b.off += p * step;
        SoaRW t = tq; t.off += p * step;
        Fp2 bx, by; bx.h = b.fp(0); by.h = b.fp(16);
        m_st_f2(t, 0, bx); m_st_f2(t, 1, by); m_st_f2(t, 2, f2_one());
    }
#pragma unroll 1
    for (int li = 0; li < ZKV_MILLER_STEPS; li++) {
        const int kind = KIND[li];
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(norm.off), "+v"(bsrc.off), "+v"(tq.off));          // see miller_loop_p
#endif
        if (kind == 0 && li != 0) f12m_sqr_body(fm);
#pragma unroll 1
        for (uint32_t p = 0; p < (uint32_t)G; p++) {
            if (!((mask >> p) & 1u)) continue;
            SoaRef n = norm; n.off += p * step;
            SoaRef b = bsrc; b.off += p * step;
            SoaRW t = tq; t.off += p * step;
            Fp2 l0, l1, l3;
            G2H T; T.x = m_ld_f2(t, 0); T.y = m_ld_f2(t, 1); T.z = m_ld_f2(t, 2);
            if (kind == 0) line_dbl(T, l0, l1, l3);
            else {
                Fp2 qx, qy; qx.h = b.fp(0); qy.h = b.fp(16);
                if (kind == 2) qy = f2_neg(qy);
                else if (kind == 3) { Fp2 x, y; g2_frob_affine(x, y, qx, qy); qx = x; qy = y; }
                else if (kind == 4) { Fp2 x, y; g2_frob2_affine(x, y, qx, qy); qx = x; qy = f2_neg(y); }
                line_add(T, qx, qy, l0, l1, l3);
            }
            m_st_f2(t, 0, T.x); m_st_f2(t, 1, T.y); m_st_f2(t, 2, T.z);
            if ((abmask >> p) & 1u) {
                const Fp2 c3 = f2_mul_fp(l1, n.fp(0)), c4 = f2_mul_fp(l3, n.fp(8));
                f12m_mul_by_034_body(fm, l0, c3, c4);
            }
        }
    }
    uint32_t fine = (1u << G) - 1u;
#pragma unroll 1
    for (uint32_t p = 0; p < (uint32_t)G; p++) {
        if (!((mask >> p) & 1u)) continue;
        SoaRef b = bsrc; b.off += p * step;
        SoaRW t = tq; t.off += p * step;
        Fp2 qx, qy; qx.h = b.fp(0); qy.h = b.fp(16);
        if (!miller_point_closes(t, qx, qy)) fine &= ~(1u << p);
    }
    return fine;
}
#endif  // ZKV_PAIRED

// ---------------------------------------------------------------- stage FINALEXP
// acc <- x^u  (acc and x are different slots; x in the cyclotomic subgroup).  Signed digits from {1, 17, 35} (gen_constants.py: the
// digit set with the fewest multiplications for this u -- a cyclotomic squaring costs well under half a multiplication): x^17 and x^35
// go to two scratch slots at W (five squarings, two multiplications), then 57 squarings and 11 multiplications, negative digits
// multiplying by the conjugate (= inverse): 13 multiplications instead of the 16 of a width-3 window and the 23 of the plain NAF.
ZKV_HD int8_t u_digit(int i) {
    const int8_t D[ZKV_U_DIG_LEN] = ZKV_U_DIG;
    return D[i];
}
template <class RA> ZKV_HD void exp_u_m(RA acc, MRef x, MRef W) {
    const MRef X17 = W, X35 = m_off(W, 96);
    f12m_copy(acc, x);
#pragma unroll 1
    for (int k = 0; k < 4; k++) f12m_cyclo_sqr(acc);   // x^16
    f12m_mul(X17, acc, x);
    f12m_copy(acc, X17); f12m_cyclo_sqr(acc);           // x^34
    f12m_mul(X35, acc, x);
    { const int t = u_digit(ZKV_U_DIG_LEN - 1); f12m_copy(acc, t == 1 ? x : t == 17 ? X17 : X35); }
#pragma unroll 1
    for (int i = ZKV_U_DIG_LEN - 2; i >= 0; i--) {
        f12m_cyclo_sqr(acc);
        const int d = u_digit(i);
        if (d == 0) continue;
        const int m = d < 0 ? -d : d;
        const MRef S = m == 1 ? x : m == 17 ? X17 : X35;
        if (d > 0) f12m_mul(acc, acc, S);
        else f12m_mul_conj(acc, acc, S);
    }
}
// f^(k (p^12-1)/r) == 1 with k = 2u(6u^2+3u+1), gcd(k, r) = 1  (Fuentes-Castaneda hard part; the chain is
// checked symbolically in tests).  F holds the Miller value on entry (clobbered); E, Y1, Y3, Y4 and the three slots at W
// are scratch; acc is the hot accumulator (LDS on the device).
template <class RA> ZKV_HD bool final_exp_is_one_m(MRef F, MRef E, MRef Y1, MRef Y3, MRef Y4, MRef W, RA acc) {
    f12m_copy(acc, F); f12m_conj(acc);
    f12m_inv(F, F);
    f12m_mul(acc, acc, F);                  // f^(p^6-1)
    f12m_frob(F, acc, 2);
    f12m_mul(E, F, acc);                    // e = ^(p^2+1)
    exp_u_m(acc, E, W); f12m_conj(acc);     // y0
    f12m_cyclo_sqr(acc); f12m_copy(Y1, acc);    // y1
    f12m_cyclo_sqr(acc);                    // y2
    f12m_mul(acc, acc, Y1); f12m_copy(Y3, acc);     // y3
    exp_u_m(acc, Y3, W); f12m_conj(acc); f12m_copy(Y4, acc);   // y4
    f12m_cyclo_sqr(acc); f12m_copy(F, acc); // y5
    exp_u_m(acc, F, W);                     // y6 (two conjugations cancel)
    f12m_conj(Y3);
    f12m_mul(acc, acc, Y4);                 // y7
    f12m_mul(acc, acc, Y3); f12m_copy(Y3, acc);     // y8
    f12m_mul(F, acc, Y1);                   // y9
    f12m_mul(acc, acc, Y4);                 // y10
    f12m_mul(acc, acc, E);                  // y11
    f12m_frob(Y1, F, 1);
    f12m_mul(acc, Y1, acc);                 // y13
    f12m_frob(Y3, Y3, 2);
    f12m_mul(acc, Y3, acc);                 // y14
    f12m_conj(E);
    f12m_mul(E, E, F);
    f12m_frob(E, E, 3);                     // y15
    f12m_mul(acc, E, acc);
    return f12m_is_one(acc);
}

#if defined(ZKV_PAIRED)
// The same final exponentiation for the lane-pair kernel as a PROGRAM of Fp12-level operations (ZKV_FE_PROG, generated from the chain
// above by gen_constants.py, with x^u on the signed digit set {1, 17, 35}) run by one loop in which every operation body is inlined
// exactly once -- no Fp12-level calls, hence no callee-saved-register frames (see miller_loop_p).
// The accumulator ACC lives in LDS in RESIDENT 29-BIT LIMBS (L9Ref; zkv_field.h "L9", zkv_tower_mem.h): the cyclotomic squaring of ACC
// (189 entries) runs on limbs throughout; ACC <- ACC * S / ACC * conj(S) with S in an HBM slot (48 entries) reads and writes ACC through
// the packing accessors; COPY to / from ACC and CONJ ACC likewise.  Every other entry works on the packed HBM slots only (the generator
// routes the few that need ACC's value through the slot TMP), in one generic body.
// slots: 0 ACC, 1 F, 2.. = E, Y1, Y3, Y4, X17, X35, TMP (consecutive 96-word slots from E).
// The HBM slots are addressed as SoaRW rows: a wave-uniform base (chosen by the program entry) and stride plus ONE 32-bit byte offset per
// lane, re-read opaquely every entry -- with per-lane 64-bit pointers the compiler computed the 48 load addresses of a multiplication
// ahead of its products and parked them in a scratch frame.  fbase / ebase: word 0 of proof 0 in the F rows / the E rows (ws.f, ws.fe);
// off = 4 (8 parity cap + proof index).
ZKV_HD SoaRW fe_slot(int s, uint32_t* fbase, uint32_t* ebase, size_t cap, uint32_t off) {
    SoaRW r; r.p = s == 1 ? fbase : ebase + (size_t)(96 * (s - 2)) * cap; r.stride = cap; r.off = off;
    return r;
}
ZKV_HD bool final_exp_prog_p(uint32_t* fbase, uint32_t* ebase, size_t cap, uint32_t off, L9Ref acc) {
    const uint32_t PROG[ZKV_FE_PROG_LEN] = ZKV_FE_PROG;
    bool one = false;
    const uint32_t slot = zkv_wave_slot_parity();
#pragma unroll 1
    for (int pc = 0; pc < ZKV_FE_PROG_LEN; pc++) {
        const uint32_t e = PROG[pc];
        const int op = (int)(e & 255u), d = (int)((e >> 8) & 255u), a = (int)((e >> 16) & 255u), b = (int)(e >> 24);
        if ((pc & 3) == 0) zkv_fair_share(slot);          // the two wavefronts of a SIMD take turns at priority (zkv_field.h)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(off));
#endif
        if (op == 6) { ZKV_MARK("begin cyclo"); f12l9_cyclo_sqr(acc); ZKV_MARK("end cyclo"); }
        else if ((op == 3 || op == 4) && d == 0) { ZKV_MARK("begin accmul");  f12l9_mul(acc, fe_slot(b, fbase, ebase, cap, off), op == 4); ZKV_MARK("end accmul"); }
        else if (op == 0 && d == 0) f12m_copy(acc, fe_slot(a, fbase, ebase, cap, off));
        else if (op == 0 && a == 0) f12m_copy(fe_slot(d, fbase, ebase, cap, off), acc);
        else if (op == 1 && d == 0) f12m_conj(acc);
        else {
            const SoaRW D = fe_slot(d, fbase, ebase, cap, off), A = fe_slot(a, fbase, ebase, cap, off);
            if (op == 3 || op == 4) { ZKV_MARK("begin genmul"); f12m_mul_body(D, A, fe_slot(b, fbase, ebase, cap, off), op == 4); ZKV_MARK("end genmul"); }
            else if (op == 0) f12m_copy(D, A);
            else if (op == 1) f12m_conj(D);
            else if (op == 2) { ZKV_MARK("begin geninv"); f12m_inv_body(D, A); ZKV_MARK("end geninv"); }
            else if (op == 5) { ZKV_MARK("begin genfrob"); f12m_frob_body(D, A, b); ZKV_MARK("end genfrob"); }
            else one = f12m_is_one(D);
        }
    }
    return one;
}
#endif

#if !defined(ZKV_PAIRED)
// ---------------------------------------------------------------- context set-up (run once per VK on the device)
ZKV_HD bool raw_g1_is_inf(const uint32_t p[2][8]) { return raw_is_zero(p[0]) && raw_is_zero(p[1]); }
ZKV_HD bool raw_g1_valid(const uint32_t p[2][8]) {            // EIP-196 input validation
    if (!raw_lt_p(p[0]) || !raw_lt_p(p[1])) return false;
    if (raw_g1_is_inf(p)) return true;
    return g1_on_curve(fp_from_raw(p[0]), fp_from_raw(p[1]));
}
ZKV_HD bool raw_g2_is_inf(const uint32_t q[4][8]) { return raw_is_zero(q[0]) && raw_is_zero(q[1]) && raw_is_zero(q[2]) && raw_is_zero(q[3]); }
ZKV_HD bool raw_g2_valid(const uint32_t q[4][8]) {            // EIP-197: on the twist and in the order-r subgroup
    for (int k = 0; k < 4; k++) if (!raw_lt_p(q[k])) return false;
    if (raw_g2_is_inf(q)) return true;
    Fp2 x, y; x.c0 = fp_from_raw(q[0]); x.c1 = fp_from_raw(q[1]); y.c0 = fp_from_raw(q[2]); y.c1 = fp_from_raw(q[3]);
    return g2_on_twist(x, y) && g2_in_subgroup(x, y);
}
// Every VK point must be an input the precompiles accept, otherwise each ecMul / ecAdd / ecPairing call of the reference
// fails and verify_proof_with_key returns false for every proof (groth16.rs:36-39, 106).
ZKV_HD void setup_validate(const VkRaw& vk, VkTables& t) {
    bool ok = raw_g1_valid(vk.alpha) && raw_g2_valid(vk.beta) && raw_g2_valid(vk.gamma) && raw_g2_valid(vk.delta);
    for (uint32_t i = 0; i < vk.n_ic; i++) ok = ok && raw_g1_valid(vk.ic[i]);
    t.vk_valid = ok ? 1u : 0u;
    t.skip_fixed[0] = raw_g2_is_inf(vk.gamma) ? 1u : 0u;
    t.skip_fixed[1] = raw_g2_is_inf(vk.delta) ? 1u : 0u;
}
ZKV_HD LineAffC line_to_table(const LineAff& l) {
    LineAffC r; r.nl.c0 = l.nl.c0; r.nl.c1 = l.nl.c1; r.c.c0 = l.c.c0; r.c.c1 = l.c.c1; return r;
}
ZKV_HD void setup_lines(const uint32_t q[4][8], LineAffC* out) {
    if (raw_g2_is_inf(q)) return;                // pair skipped (skip_fixed)
    G2A Q, T;
    Q.x.c0 = fp_from_raw(q[0]); Q.x.c1 = fp_from_raw(q[1]); Q.y.c0 = fp_from_raw(q[2]); Q.y.c1 = fp_from_raw(q[3]);
    T = Q;
    Fp2 nqy = f2_neg(Q.y);
    int li = 0;
#pragma unroll 1
    for (int i = ZKV_ATE_NAF_LEN - 2; i >= 0; i--) {
        out[li++] = line_to_table(aff_dbl(T));
        int d = ate_naf(i);
        if (d != 0) out[li++] = line_to_table(aff_add(T, Q.x, d > 0 ? Q.y : nqy));
    }
    Fp2 q1x, q1y, q2x, q2y;
    g2_frob_affine(q1x, q1y, Q.x, Q.y);
    g2_frob2_affine(q2x, q2y, Q.x, Q.y);
    out[li++] = line_to_table(aff_add(T, q1x, q1y));
    out[li++] = line_to_table(aff_add(T, q2x, f2_neg(q2y)));
}
ZKV_HD G1J g1_mul_raw(const Fp& x, const Fp& y, const uint32_t k[8]) {
    G1J acc = g1j_infinity();
#pragma unroll 1
    for (int i = 255; i >= 0; i--) {
        acc = g1j_dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1u) acc = g1j_add_affine(acc, x, y);
    }
    return acc;
}
ZKV_HD void g1j_to_affine(const G1J& p, G1A& out, uint32_t& inf) {
    if (fp_is_zero(p.z)) { inf = 1; out.x = fp_zero(); out.y = fp_zero(); return; }
    Fp zi = fp_inv(p.z), zi2 = fp_sqr(zi);
    out.x = fp_mul(p.x, zi2); out.y = fp_mul(p.y, fp_mul(zi2, zi)); inf = 0;
}
// base = IC[0] + sum over fixed signals s_i * IC[i]
ZKV_HD void fixed_signal_base(const VkRaw& vk, const uint32_t scalar[MAX_IC][8], G1A& base, uint32_t& base_inf) {
    G1J acc;
    if (raw_g1_is_inf(vk.ic[0])) acc = g1j_infinity();
    else { acc.x = fp_from_raw(vk.ic[0][0]); acc.y = fp_from_raw(vk.ic[0][1]); acc.z = fp_one(); }
#pragma unroll 1
    for (uint32_t i = 1; i < vk.n_ic; i++) {
        if (!vk.is_fixed[i] || raw_g1_is_inf(vk.ic[i])) continue;
        G1J m = g1_mul_raw(fp_from_raw(vk.ic[i][0]), fp_from_raw(vk.ic[i][1]), scalar[i]);
        G1A ma; uint32_t inf;
        g1j_to_affine(m, ma, inf);
        if (!inf) acc = g1j_add_affine(acc, ma.x, ma.y);
    }
    g1j_to_affine(acc, base, base_inf);
}
ZKV_HD void setup_base(const VkRaw& vk, VkTables& t) {
    fixed_signal_base(vk, vk.fixed_scalar, t.base, t.base_inf);
    t.n_var = vk.n_var;
    for (int b = 0; b < MAX_VAR; b++) t.var_windows[b] = vk.var_windows[b];
}
// one window row of a fixed-base table: d * 256^w * P, d = 1..255, all affine (P = (x, y), not infinity).
// The row is built by doubling levels (1 | 2 3 | 4..7 | ... | 128..255): level m adds mP to the known multiples 1..m-1 and doubles
// mP; the m slopes of a level share ONE field inversion (prefix products parked in the y slots that are about to be written).
ZKV_HD void setup_window_row(const Fp& x, const Fp& y, int w, G1A* row) {
    G1J p; p.x = x; p.y = y; p.z = fp_one();
#pragma unroll 1
    for (int i = 0; i < 8 * w; i++) p = g1j_dbl(p);
    Fp zi = fp_inv(p.z), zi2 = fp_sqr(zi);
    row[0].x = fp_zero(); row[0].y = fp_zero();
    row[1].x = fp_mul(p.x, zi2); row[1].y = fp_mul(p.y, fp_mul(zi2, zi));
#pragma unroll 1
    for (int m = 1; m < MSM_DIGITS; m *= 2) {
        const Fp xm = row[m].x, ym = row[m].y;
        const int cnt = (2 * m < MSM_DIGITS) ? m : m - 1;       // the last level stops at 255
        Fp prod = fp_one();
#pragma unroll 1
        for (int j = 1; j <= cnt; j++) {
            Fp den = (j < m) ? fp_sub(row[j].x, xm) : fp_dbl(ym);
            row[m + j].y = prod;
            prod = fp_mul(prod, den);
        }
        Fp inv = fp_inv(prod);
#pragma unroll 1
        for (int j = cnt; j >= 1; j--) {
            Fp den = (j < m) ? fp_sub(row[j].x, xm) : fp_dbl(ym);
            Fp dinv = fp_mul(inv, row[m + j].y);
            inv = fp_mul(inv, den);
            Fp xj = (j < m) ? row[j].x : xm;
            Fp num = (j < m) ? fp_sub(row[j].y, ym) : fp_add(fp_dbl(fp_sqr(xm)), fp_sqr(xm));
            Fp lam = fp_mul(num, dinv);
            Fp x3 = fp_sub(fp_sub(fp_sqr(lam), xm), xj);
            row[m + j].x = x3;
            row[m + j].y = fp_sub(fp_mul(lam, fp_sub(xm, x3)), ym);
        }
    }
}
// one (b, w) row of the vk_x table: d * 256^w * IC_var[b].  Rows w >= var_windows[b] are never read by the vk_x stage (a 128-bit signal
// has 16 windows); the aggregate check multiplies IC_var[b] by full-width sums and reads all 32.
ZKV_HD void setup_msm_row(const VkRaw& vk, VkTables& t, int b, int w) {
    uint32_t ici = vk.var_ic[b];
    if (raw_g1_is_inf(vk.ic[ici])) {             // s * infinity = infinity: leave the row zero and never read it
        if (w == 0) t.var_windows[b] = 0;
        return;
    }
    setup_window_row(fp_from_raw(vk.ic[ici][0]), fp_from_raw(vk.ic[ici][1]), w, t.msm[b][w]);
}
// 64 entries of one 16-bit row: d = 256 hi + lo, lo = lo0 .. lo0 + 63, as (lo * 256^(2w) + hi * 256^(2w+1)) * IC_var[b] = the sum of one
// entry of each of the two 8-bit rows the window covers.  The 64 chords share one field inversion (prefix products parked in the y
// slots about to be written).  Neither the two points nor their negatives ever coincide: lo < 256 <= 256 hi and 256 hi + lo < r.
ZKV_HD void setup_msm16_chunk(const VkTables& t, G1A* row, uint32_t b, uint32_t w, uint32_t hi, uint32_t lo0) {
    const uint32_t nw = t.var_windows[b];
    if (2 * w >= nw) return;                                    // also IC_b = infinity (var_windows 0): the row is never read
    const G1A* L = t.msm[b][2 * w];
    G1A* out = row + hi * 256 + lo0;
    if (hi == 0) {
#pragma unroll 1
        for (uint32_t j = 0; j < 64; j++) out[j] = L[lo0 + j];
        return;
    }
    if (2 * w + 1 >= nw) return;                                // no upper half: digits above 255 do not occur
    const G1A H = t.msm[b][2 * w + 1][hi];
    Fp prod = fp_one();
#pragma unroll 1
    for (uint32_t j = 0; j < 64; j++) {
        if (lo0 + j == 0) continue;
        out[j].y = prod;
        prod = fp_mul(prod, fp_sub(L[lo0 + j].x, H.x));
    }
    Fp inv = fp_inv(prod);
#pragma unroll 1
    for (int j = 63; j >= 0; j--) {
        if (lo0 + j == 0) { out[j] = H; continue; }
        const G1A P = L[lo0 + j];
        const Fp dinv = fp_mul(inv, out[j].y);
        inv = fp_mul(inv, fp_sub(P.x, H.x));
        const Fp lam = fp_mul(fp_sub(P.y, H.y), dinv);
        const Fp x3 = fp_sub(fp_sub(fp_sqr(lam), H.x), P.x);
        out[j].x = x3;
        out[j].y = fp_sub(fp_mul(lam, fp_sub(H.x, x3)), H.y);
    }
}
// One instance of a RISC Zero verifier set: what `initialize` derives from (control_root, bn254_control_id)
// (risc0/verifier.rs:58-76) -- the selector (verifier.rs:128-144: tagged SHA-256 over control root, byte-reversed control
// id and the VK digest), the split control root (crypto.rs:95-110) -- plus this library's per-instance part of vk_x.
ZKV_HD void setup_instance(const VkRaw& vk, const InstConsts& k, const InstRaw& in, InstTab& out) {
    uint8_t buf[130];
    for (int i = 0; i < 32; i++) { buf[i] = k.tag[i]; buf[32 + i] = in.control_root[i]; buf[64 + i] = in.control_id[31 - i]; buf[96 + i] = k.vk_digest[i]; }
    buf[128] = 3; buf[129] = 0;
    uint32_t h[8];
    sha256_bytes(buf, 130, h);
    out.selector_be = h[0];
    uint32_t sc[MAX_IC][8];
    for (int i = 0; i < MAX_IC; i++) for (int j = 0; j < 8; j++) sc[i][j] = 0;
    // split_digest: byte-reverse the root; low = rev[16..32], high = rev[0..16], each read big-endian as a 128-bit scalar
    for (int j = 0; j < 4; j++) {
        sc[1][j] = __builtin_bswap32(load_be32(in.control_root + 4 * j));            // control_root_0
        sc[2][j] = __builtin_bswap32(load_be32(in.control_root + 16 + 4 * j));       // control_root_1
    }
    load_be256(sc[5], in.control_id);
    out.fail = raw_lt_r(sc[5]) ? 0u : 1u;              // a control id >= R fails every proof at groth16.rs:32
    if (out.fail) for (int j = 0; j < 8; j++) sc[5][j] = 0;
    fixed_signal_base(vk, sc, out.base, out.base_inf);
    out.pad = 0;
}
ZKV_HD void setup_alpha_beta(const VkRaw& vk, VkTables& t, MRef fm, MRef tm) {
    if (raw_g1_is_inf(vk.alpha) || raw_g2_is_inf(vk.beta)) {       // e(alpha, beta) contributes 1
        f12m_set_one(fm);
        for (int k = 0; k < 96; k++) t.f_alpha_beta[k] = fm.p[(size_t)k * fm.stride];
        return;
    }
    Fp ax = fp_from_raw(vk.alpha[0]), ay = fp_from_raw(vk.alpha[1]);
    Fp iy = fp_inv(ay);
    G1Norm n; n.axs = fp_mul(ax, iy); n.ays = iy;
    n.lxs = n.lys = n.cxs = n.cys = fp_zero();
    Fp2 bx, by;
    bx.c0 = fp_from_raw(vk.beta[0]); bx.c1 = fp_from_raw(vk.beta[1]);
    by.c0 = fp_from_raw(vk.beta[2]); by.c1 = fp_from_raw(vk.beta[3]);
    miller_loop_m((const VkTables*)nullptr, 0, n, bx, by, fm, tm);
    for (int k = 0; k < 96; k++) t.f_alpha_beta[k] = fm.p[(size_t)k * fm.stride];
}

#endif  // !ZKV_PAIRED

}  // namespace zkv
