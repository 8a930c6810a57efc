// Host build of the kernel math (stylus_zkvm_verifiers_amd/csrc/zkv_*.h) for CPU-side tests.
// TEST ONLY: the shipped library never runs these stages on the host; this file lets `-m "not gpu"` tests
// check the exact stage functions the HIP kernels execute against the oracle and the golden vectors.
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#define ZKV_COUNT_FP_MUL 1
#include "../../stylus_zkvm_verifiers_amd/csrc/zkv_host_vk.h"

using namespace zkv;

static VkTables* g_tab[2] = {nullptr, nullptr};
static uint8_t g_key[2][64];

static VkTables* tables(int vm, const uint8_t* cr, const uint8_t* cid) {
    uint8_t key[64]; memset(key, 0, 64);
    if (vm == 0) { memcpy(key, cr, 32); memcpy(key + 32, cid, 32); }
    if (g_tab[vm] && memcmp(key, g_key[vm], 64) == 0) return g_tab[vm];
    VkRaw raw;
    if (vm == 0) { uint8_t lo[16], hi[16]; host::split_digest(cr, lo, hi); host::fill_vk_risc0(raw, lo, hi, cid); }
    else host::fill_vk_sp1(raw);
    VkTables* t = g_tab[vm] ? g_tab[vm] : (VkTables*)malloc(sizeof(VkTables));
    memset(t, 0, sizeof *t);
    setup_base(raw, *t);
    setup_validate(raw, *t);
    for (uint32_t b = 0; b < raw.n_var; b++) for (uint32_t w = 0; w < raw.var_windows[b]; w++) setup_msm_row(raw, *t, (int)b, (int)w);
    setup_lines(raw.gamma, t->lines[0]);
    setup_lines(raw.delta, t->lines[1]);
    { uint32_t ab[96 + 48]; MRef fm = m_ref(ab, 1), tm = m_ref(ab + 96, 1); setup_alpha_beta(raw, *t, fm, tm); }
    g_tab[vm] = t; memcpy(g_key[vm], key, 64);
    return t;
}

static unsigned long long g_stage_muls[5], g_stage_mads[5];

// The 16-bit window rows (Msm16) for the scalars of ONE proof: only the 64-entry chunks its digits touch are built, in a lazily zeroed
// allocation of the table's full size.  The caller frees m.tab.
static Msm16 msm16_for(const VkTables& t, const PrepOut& p) {
    Msm16 m = {nullptr, {0, 0, 0, 0, 0}};
    uint32_t rows = 0;
    for (uint32_t b = 0; b < t.n_var; b++) { m.row0[b] = rows; rows += (t.var_windows[b] + 1) / 2; }
    G1A* tab = (G1A*)calloc((size_t)rows << 16, sizeof(G1A));
    for (uint32_t b = 0; b < t.n_var; b++)
        for (uint32_t w = 0; 2 * w < t.var_windows[b]; w++) {
            const uint32_t d = (p.s[b][w >> 1] >> ((w & 1) * 16)) & 0xffffu;
            setup_msm16_chunk(t, tab + ((size_t)(m.row0[b] + w) << 16), b, w, d >> 8, d & 0xc0u);
        }
    m.tab = tab;
    return m;
}
extern "C" {
// Montgomery multiplications spent in each stage (prep, msm, g2chk, miller, finalexp) by the last hs_groth16 call
void hs_stage_muls(unsigned long long* out) { for (int i = 0; i < 5; i++) out[i] = g_stage_muls[i]; }
// ... and the 32 x 32 + 64 multiply-adds (v_mad_u64_u32 on the device) those multiplications are made of
void hs_stage_mads(unsigned long long* out) { for (int i = 0; i < 5; i++) out[i] = g_stage_mads[i]; }

// returns 1 accept / 0 reject for the Groth16 core given the 8 proof words and the two per-proof scalars
int hs_groth16(int vm, const uint8_t* cr, const uint8_t* cid, const uint8_t* words, const uint8_t* s0, const uint8_t* s1) {
    VkTables* t = tables(vm, cr, cid);
    if (vm == 0) { uint32_t id[8]; load_be256(id, cid); if (!raw_lt_r(id)) return 0; }
    uint32_t w[8][8];
    for (int k = 0; k < 8; k++) load_be256(w[k], words + 32 * k);
    PrepOut p; memset(&p, 0, sizeof p);
    load_be256(p.s[0], s0); load_be256(p.s[1], s1);
    if (!raw_lt_r(p.s[0]) || !raw_lt_r(p.s[1])) return 0;
    unsigned long long c0 = zkv_fp_mul_counter, d0 = zkv_mad_counter;
    if (!prep_points(w, vm == 0, p)) return 0;
    g_stage_muls[0] = zkv_fp_mul_counter - c0; c0 = zkv_fp_mul_counter; g_stage_mads[0] = zkv_mad_counter - d0; d0 = zkv_mad_counter;
    if (!(p.flags & FL_B_INF) && !g2_in_subgroup(p.bx, p.by)) return 0;
    g_stage_muls[2] = zkv_fp_mul_counter - c0; c0 = zkv_fp_mul_counter; g_stage_mads[2] = zkv_mad_counter - d0; d0 = zkv_mad_counter;
    // the vk_x stage as big batches run it (k_msm on the 16-bit window rows); building this proof's part of the rows is set-up, not counted
    G1Norm n; uint32_t fl = p.flags;
    const Msm16 m16 = msm16_for(*t, p);
    c0 = zkv_fp_mul_counter; d0 = zkv_mad_counter;
    msm_normalize_acc(msm_accumulate_w16(*t, m16, [&](uint32_t b, uint32_t k) { return p.s[b][k]; }, t->base, t->base_inf), p, fl, n);
    free((void*)m16.tab);
    g_stage_muls[1] = zkv_fp_mul_counter - c0; c0 = zkv_fp_mul_counter; g_stage_mads[1] = zkv_mad_counter - d0; d0 = zkv_mad_counter;
    // same slot structure as the kernels: f and T in one buffer (LDS on the device), 5 Fp12 slots for the final exp
    static thread_local uint32_t buf[96 + 48], slots[8 * 96];
    MRef fm = m_ref(buf, 1), tm = m_ref(buf + 96, 1);
    if (!miller_loop_m(t, fl, n, p.bx, p.by, fm, tm, true)) return -1;     // the loop's own subgroup verdict must agree with the classical test above
    g_stage_muls[3] = zkv_fp_mul_counter - c0; c0 = zkv_fp_mul_counter; g_stage_mads[3] = zkv_mad_counter - d0; d0 = zkv_mad_counter;
    MRef F = m_ref(slots, 1), E = m_ref(slots + 96, 1), Y1 = m_ref(slots + 192, 1), Y3 = m_ref(slots + 288, 1), Y4 = m_ref(slots + 384, 1);
    for (int k = 0; k < 96; k++) slots[k] = t->f_alpha_beta[k];
    f12m_mul(F, F, fm);
    int acc = final_exp_is_one_m(F, E, Y1, Y3, Y4, m_ref(slots + 480, 1), fm) ? 1 : 0;
    g_stage_muls[4] = zkv_fp_mul_counter - c0; g_stage_mads[4] = zkv_mad_counter - d0;
    return acc;
}
// Runs PREP + MSM for one proof and exports what the Fp2-heavy stages consume (for the lane-pair host emulation,
// host_sim_paired.cpp).  Returns the device-table pointer, or NULL when the proof is rejected before the pairing.
const void* hs_prepare(int vm, const uint8_t* cr, const uint8_t* cid, const uint8_t* words, const uint8_t* s0, const uint8_t* s1,
                       uint32_t* flags_out, uint32_t* norm48, uint32_t* b32) {
    VkTables* t = tables(vm, cr, cid);
    if (vm == 0) { uint32_t id[8]; load_be256(id, cid); if (!raw_lt_r(id)) return nullptr; }
    uint32_t w[8][8];
    for (int k = 0; k < 8; k++) load_be256(w[k], words + 32 * k);
    PrepOut p; memset(&p, 0, sizeof p);
    load_be256(p.s[0], s0); load_be256(p.s[1], s1);
    if (!raw_lt_r(p.s[0]) || !raw_lt_r(p.s[1])) return nullptr;
    if (!prep_points(w, vm == 0, p)) return nullptr;
    G1Norm n; uint32_t fl = p.flags;
    msm_normalize(*t, p, fl, n);
    *flags_out = fl;
    const Fp* nf[6] = {&n.axs, &n.ays, &n.lxs, &n.lys, &n.cxs, &n.cys};
    for (int k = 0; k < 6; k++) memcpy(norm48 + 8 * k, nf[k]->v, 32);
    memcpy(b32, p.bx.c0.v, 32); memcpy(b32 + 8, p.bx.c1.v, 32); memcpy(b32 + 16, p.by.c0.v, 32); memcpy(b32 + 24, p.by.c1.v, 32);
    return t;
}
// compute_vk_x through the windowed fixed-base tables (the k_msm / k_vk_x code), affine result as 64 big-endian bytes
void hs_vk_x(int vm, const uint8_t* cr, const uint8_t* cid, const uint8_t* s0, const uint8_t* s1, uint8_t* out64) {
    VkTables* t = tables(vm, cr, cid);
    PrepOut p; memset(&p, 0, sizeof p);
    load_be256(p.s[0], s0); load_be256(p.s[1], s1);
    G1J acc = msm_accumulate(*t, p);
    G1A a; uint32_t inf; g1j_to_affine(acc, a, inf);
    uint32_t r[8];
    for (int c = 0; c < 2; c++) {
        fp_to_raw(r, c ? a.y : a.x);
        for (int i = 0; i < 8; i++) for (int k = 0; k < 4; k++) out64[32 * c + 31 - 4 * i - k] = r[i] >> (8 * k);
    }
}
// The same through the 16-bit window rows (Msm16: setup_msm16_chunk + msm_accumulate_w16, what k_setup_msm16 / k_msm run for big batches)
void hs_vk_x16(int vm, const uint8_t* cr, const uint8_t* cid, const uint8_t* s0, const uint8_t* s1, uint8_t* out64) {
    VkTables* t = tables(vm, cr, cid);
    PrepOut p; memset(&p, 0, sizeof p);
    load_be256(p.s[0], s0); load_be256(p.s[1], s1);
    const Msm16 m = msm16_for(*t, p);
    G1J acc = msm_accumulate_w16(*t, m, [&](uint32_t b, uint32_t k) { return p.s[b][k]; }, t->base, t->base_inf);
    free((void*)m.tab);
    G1A a; uint32_t inf; g1j_to_affine(acc, a, inf);
    uint32_t r[8];
    for (int c = 0; c < 2; c++) {
        fp_to_raw(r, c ? a.y : a.x);
        for (int i = 0; i < 8; i++) for (int k = 0; k < 4; k++) out64[32 * c + 31 - 4 * i - k] = r[i] >> (8 * k);
    }
}
// verify_proof_with_key for an arbitrary key through the kernel stage functions (tables rebuilt per call: test only)
int hs_groth16_generic(const uint8_t* vk_words, int n_ic, int negate_a, const uint8_t* words, const uint8_t* signals) {
    static VkTables* t = (VkTables*)malloc(sizeof(VkTables));
    VkRaw raw; host::fill_vk_generic(raw, vk_words, (uint32_t)n_ic);
    memset(t, 0, sizeof *t);
    setup_validate(raw, *t);
    if (!t->vk_valid) return 0;
    setup_base(raw, *t);
    for (uint32_t b = 0; b < raw.n_var; b++) for (uint32_t w = 0; w < raw.var_windows[b]; w++) setup_msm_row(raw, *t, (int)b, (int)w);
    setup_lines(raw.gamma, t->lines[0]);
    setup_lines(raw.delta, t->lines[1]);
    { uint32_t ab[96 + 48]; MRef fm = m_ref(ab, 1), tm = m_ref(ab + 96, 1); setup_alpha_beta(raw, *t, fm, tm); }
    PrepOut p; memset(&p, 0, sizeof p);
    for (int b = 0; b + 1 < n_ic; b++) { load_be256(p.s[b], signals + 32 * b); if (!raw_lt_r(p.s[b])) return 0; }
    uint32_t w[8][8];
    for (int k = 0; k < 8; k++) load_be256(w[k], words + 32 * k);
    if (!prep_points(w, negate_a != 0, p)) return 0;
    if (!(p.flags & FL_B_INF) && !g2_in_subgroup(p.bx, p.by)) return 0;
    G1Norm n; uint32_t fl = p.flags;
    msm_normalize(*t, p, fl, n);
    static thread_local uint32_t buf[96 + 48], slots[8 * 96];
    MRef fm = m_ref(buf, 1), tm = m_ref(buf + 96, 1);
    miller_loop_m(t, fl, n, p.bx, p.by, fm, tm);
    MRef F = m_ref(slots, 1), E = m_ref(slots + 96, 1), Y1 = m_ref(slots + 192, 1), Y3 = m_ref(slots + 288, 1), Y4 = m_ref(slots + 384, 1);
    for (int k = 0; k < 96; k++) slots[k] = t->f_alpha_beta[k];
    f12m_mul(F, F, fm);
    return final_exp_is_one_m(F, E, Y1, Y3, Y4, m_ref(slots + 480, 1), fm) ? 1 : 0;
}
void hs_risc0_scalars(const uint8_t* image_id, const uint8_t* journal, uint8_t* digest32, uint8_t* lo32, uint8_t* hi32) {
    Risc0Consts k; host::risc0_consts(k);
    uint32_t h[8], lo[8], hi[8];
    risc0_claim_digest(k, image_id, journal, h);
    for (int i = 0; i < 8; i++) { digest32[4 * i] = h[i] >> 24; digest32[4 * i + 1] = h[i] >> 16; digest32[4 * i + 2] = h[i] >> 8; digest32[4 * i + 3] = h[i]; }
    risc0_split_digest(h, lo, hi);
    for (int i = 0; i < 8; i++) for (int b = 0; b < 4; b++) { lo32[31 - 4 * i - b] = lo[i] >> (8 * b); hi32[31 - 4 * i - b] = hi[i] >> (8 * b); }
}
void hs_sha256(const uint8_t* m, size_t n, uint8_t* out) { host::sha256_host(m, n, out); }
void hs_risc0_selector(const uint8_t* cr, const uint8_t* cid, uint8_t* sel4, uint8_t* vkd32) { host::risc0_selector(cr, cid, sel4); host::risc0_vk_digest(vkd32); }
// canonical big-endian a*b mod p through the Montgomery kernels' code
void hs_fp_mulmod(const uint8_t* a, const uint8_t* b, uint8_t* out) {
    uint32_t x[8], y[8], r[8];
    load_be256(x, a); load_be256(y, b);
    fp_to_raw(r, fp_mul(fp_from_raw(x), fp_from_raw(y)));
    for (int i = 0; i < 8; i++) for (int k = 0; k < 4; k++) out[31 - 4 * i - k] = r[i] >> (8 * k);
}
// Loose-range invariance: every field operation must give the same residue whichever representation (x or x + p) its
// operands use, results must stay below 2p, and the comparisons must treat both representations as equal.
// a, b: canonical big-endian values.  Returns 0 when everything holds, otherwise the number of the failing check.
static Fp plus_p(const Fp& x) { const uint32_t P[8] = ZKV_FP_P_LIMBS; Fp r; uint32_t c = 0; for (int i = 0; i < 8; i++) r.v[i] = addc(x.v[i], P[i], c); return r; }
static bool below_2p(const Fp& x) { const uint32_t P2[8] = ZKV_FP_2P_LIMBS; return !u256_geq(x.v, P2); }
static bool same_residue(const Fp& x, const Fp& y) { uint32_t a[8], b[8]; fp_to_raw(a, x); fp_to_raw(b, y); for (int i = 0; i < 8; i++) if (a[i] != b[i]) return false; return true; }
int hs_fp_loose_check(const uint8_t* a32, const uint8_t* b32) {
    uint32_t ra[8], rb[8];
    load_be256(ra, a32); load_be256(rb, b32);
    Fp A[2], B[2];
    A[0] = fp_from_raw(ra); B[0] = fp_from_raw(rb);
    // representatives as produced by the multiplier, then their + p twins
    A[1] = plus_p(A[0]); B[1] = plus_p(B[0]);
    if (!below_2p(A[0]) || !below_2p(B[0])) return 1;
    if (!below_2p(A[1])) A[1] = A[0];                // A[0] was already >= p: its twin would leave the range
    if (!below_2p(B[1])) B[1] = B[0];
    Fp ref_add = fp_add(A[0], B[0]), ref_sub = fp_sub(A[0], B[0]), ref_mul = fp_mul(A[0], B[0]), ref_neg = fp_neg(A[0]);
    Fp ref_lazy = fp_mul(fp_add_nr(A[0], B[0]), fp_add_nr(B[0], B[0]));
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) {
        Fp s = fp_add(A[i], B[j]), d = fp_sub(A[i], B[j]), m = fp_mul(A[i], B[j]), n = fp_neg(A[i]);
        Fp lz = fp_mul(fp_add_nr(A[i], B[j]), fp_add_nr(B[j], B[1 - j]));
        if (!below_2p(s) || !below_2p(d) || !below_2p(m) || !below_2p(n) || !below_2p(lz)) return 2;
        if (!same_residue(s, ref_add) || !fp_eq(s, ref_add)) return 3;
        if (!same_residue(d, ref_sub) || !fp_eq(d, ref_sub)) return 4;
        if (!same_residue(m, ref_mul) || !fp_eq(m, ref_mul)) return 5;
        if (!same_residue(n, ref_neg)) return 6;
        if (!same_residue(lz, ref_lazy)) return 7;
        if (!fp_eq(A[i], A[1 - i]) || !fp_is_zero(fp_sub(A[i], A[1 - i]))) return 8;
        { Fp hf = fp_half(A[i]); if (!below_2p(hf) || !same_residue(fp_add(hf, hf), A[0])) return 16; }
        Fp x2[2], y2[2], r2[2]; x2[0] = A[i]; x2[1] = B[j]; y2[0] = B[j]; y2[1] = A[i];
        fp_add_n<2>(x2, y2, r2);
        if (!same_residue(r2[0], ref_add) || !below_2p(r2[1])) return 9;
        fp_sub_n<2>(x2, y2, r2);
        if (!same_residue(r2[0], ref_sub) || !below_2p(r2[0]) || !below_2p(r2[1])) return 10;
        Fp r0, r1;
        fp_add_x2(A[i], B[j], B[j], A[i], r0, r1);
        if (!same_residue(r0, ref_add) || !same_residue(r1, ref_add) || !below_2p(r0)) return 11;
        fp_sub_x2(A[i], B[j], B[j], A[i], r0, r1);
        if (!same_residue(r0, ref_sub) || !below_2p(r0) || !below_2p(r1)) return 12;
    }
    bool az = true; for (int i = 0; i < 8; i++) az = az && ra[i] == 0;
    if (fp_is_zero(A[0]) != az || fp_is_zero(A[1]) != az) return 13;
    if (same_residue(A[0], B[0]) != fp_eq(A[1], B[0])) return 14;
    Fp inv = fp_inv(A[1]);
    if (!az && !same_residue(fp_mul(inv, A[0]), fp_one())) return 15;
    return 0;
}
// One-proof-per-lane multipliers on raw operands below 4p (the edge of their contract): out = a*b (Fp, canonical) followed by
// the two components of (a0 + a1 u)(b0 + b1 u); returns 0 when a result leaves the loose range.
int hs_mul_edge(const uint32_t* in32, uint32_t* out24) {
    const uint32_t P2[8] = ZKV_FP_2P_LIMBS;
    Fp a0, a1, b0, b1;
    memcpy(a0.v, in32, 32); memcpy(a1.v, in32 + 8, 32); memcpy(b0.v, in32 + 16, 32); memcpy(b1.v, in32 + 24, 32);
    Fp m = fp_mul(a0, b0);
    Fp2 x, y; x.c0 = a0; x.c1 = a1; y.c0 = b0; y.c1 = b1;
    Fp2 z = f2_mul(x, y);
    if (u256_geq(m.v, P2) || u256_geq(z.c0.v, P2) || u256_geq(z.c1.v, P2)) return 0;
    fp_to_raw(out24, m); fp_to_raw(out24 + 8, z.c0); fp_to_raw(out24 + 16, z.c1);
    return 1;
}
// returns 1 when the Granger-Scott cyclotomic squaring equals the generic squaring on a random element of the
// cyclotomic subgroup (x^((p^6-1)(p^2+1)) for x built from the seed bytes)
int hs_cyclo_sqr_check(const uint8_t* seed384) {
    uint32_t buf[4 * 96];
    MRef X = m_ref(buf, 1), A = m_ref(buf + 96, 1), B = m_ref(buf + 192, 1), C = m_ref(buf + 288, 1);
    for (int k = 0; k < 12; k++) { uint32_t w[8]; load_be256(w, seed384 + 32 * k); w[7] &= 0x0fffffffu; m_st_fp(X, 8 * k, fp_from_raw(w)); }
    f12m_copy(A, X); f12m_conj(A); f12m_inv(B, X); f12m_mul(A, A, B);      // ^(p^6-1)
    f12m_frob(B, A, 2); f12m_mul(A, B, A);                                  // ^(p^2+1)
    f12m_copy(B, A); f12m_copy(C, A);
    f12m_sqr(B); f12m_cyclo_sqr(C);
    for (int k = 0; k < 12; k++) if (!fp_eq(m_ld_fp(B, 8 * k), m_ld_fp(C, 8 * k))) return 0;      // values live in the loose range [0, 2p)
    return f12m_is_one(A) ? -1 : 1;
}
// One instance of a verifier set through setup_instance (the code of k_setup_instances): selector, control-id range flag and
// the instance's vk_x for the given claim halves (64 bytes x, y; all-zero = infinity) through the windowed MSM.
int hs_set_instance(const uint8_t* cr, const uint8_t* cid, const uint8_t* s0, const uint8_t* s1, uint8_t* sel4, uint8_t* vkx64) {
    const uint8_t zero[32] = {0};
    VkTables* t = tables(0, zero, zero);            // VK-level tables; the context's own fixed signals are not used
    VkRaw raw; { uint8_t lo[16] = {0}, hi[16] = {0}; host::fill_vk_risc0(raw, lo, hi, zero); }
    InstConsts k;
    host::sha256_host((const uint8_t*)"risc0.Groth16ReceiptVerifierParameters", 38, k.tag);
    host::risc0_vk_digest(k.vk_digest);
    InstRaw in; memcpy(in.control_root, cr, 32); memcpy(in.control_id, cid, 32);
    InstTab tab;
    setup_instance(raw, k, in, tab);
    for (int b = 0; b < 4; b++) sel4[b] = (uint8_t)(tab.selector_be >> (24 - 8 * b));
    PrepOut p; memset(&p, 0, sizeof p);
    load_be256(p.s[0], s0); load_be256(p.s[1], s1);
    G1J acc = msm_accumulate(*t, p, tab.base, tab.base_inf);
    G1A a; uint32_t inf;
    g1j_to_affine(acc, a, inf);
    uint32_t r[8];
    for (int c = 0; c < 2; c++) {
        fp_to_raw(r, c ? a.y : a.x);
        for (int i = 0; i < 8; i++) for (int b = 0; b < 4; b++) vkx64[32 * c + 31 - 4 * i - b] = (uint8_t)(r[i] >> (8 * b));
    }
    return (int)tab.fail;
}
// The subgroup verdict the Miller loop produces on its way (miller_loop_m / miller_loop_p with check_b), for an on-twist point alone:
// no G1 point, no line products, only the running point.
int hs_g2_in_subgroup_by_miller(const uint8_t* q128) {
    uint32_t w[4][8];
    for (int k = 0; k < 4; k++) load_be256(w[k], q128 + 32 * k);
    Fp2 x, y; x.c1 = fp_from_raw(w[0]); x.c0 = fp_from_raw(w[1]); y.c1 = fp_from_raw(w[2]); y.c0 = fp_from_raw(w[3]);
    if (!g2_on_twist(x, y)) return -1;
    static thread_local uint32_t buf[96 + 48];
    G1Norm n; n.axs = n.ays = n.lxs = n.lys = n.cxs = n.cys = fp_zero();
    return miller_loop_m((const VkTables*)nullptr, FL_A_INF, n, x, y, m_ref(buf, 1), m_ref(buf + 96, 1), true) ? 1 : 0;
}
// Exceptional cases of the incomplete tangent / chord formulas (T = B, T = -B, T = O): each must leave Z = 0, and Z = 0 must
// survive further tangent and chord steps -- what the Miller-loop subgroup verdict relies on for points outside G2.
// Returns a bit mask of the checks that FAILED (0 = all hold).
int hs_line_exceptional(const uint8_t* q128) {
    uint32_t w[4][8];
    for (int k = 0; k < 4; k++) load_be256(w[k], q128 + 32 * k);
    Fp2 x, y; x.c1 = fp_from_raw(w[0]); x.c0 = fp_from_raw(w[1]); y.c1 = fp_from_raw(w[2]); y.c0 = fp_from_raw(w[3]);
    int bad = 0;
    Fp2 l0, l1, l3;
    for (int c = 0; c < 3; c++) {
        G2H T; T.x = x; T.y = c == 1 ? f2_neg(y) : y; T.z = f2_one();
        if (c == 2) { T.x = f2_zero(); T.z = f2_zero(); }            // O = (0 : Y : 0)
        // scale to a non-trivial projective representative
        const Fp2 k = f2_add(f2_mul(x, y), f2_one());
        T.x = f2_mul(T.x, k); T.y = f2_mul(T.y, k); T.z = f2_mul(T.z, k);
        line_add(T, x, y, l0, l1, l3);
        if (!f2_is_zero(T.z)) bad |= 1 << c;
        line_dbl(T, l0, l1, l3);
        if (!f2_is_zero(T.z)) bad |= 8 << c;
        line_add(T, x, y, l0, l1, l3);
        if (!f2_is_zero(T.z)) bad |= 64 << c;
    }
    { G2H T; T.x = x; T.y = f2_zero(); T.z = f2_one(); line_dbl(T, l0, l1, l3); if (!f2_is_zero(T.z)) bad |= 512; }   // a 2-torsion shape (Y = 0)
    return bad;
}
// miller_point_closes on hand-made running points for B = q (EIP-197 word order): bit 0 = verdict for T = -psi^3(B) scaled by a
// non-trivial Z (must be 1), bit 1 = the same coordinates with Z = 0 (must be 0: an exceptional case of the incomplete formulas
// leaves Z = 0, and the closing test is what rejects it), bit 2 = T = (0 : Y : 0) (must be 0), bit 3 = T = +psi^3(B) (must be 0).
int hs_miller_closing_test(const uint8_t* q128) {
    uint32_t w[4][8];
    for (int k = 0; k < 4; k++) load_be256(w[k], q128 + 32 * k);
    Fp2 x, y; x.c1 = fp_from_raw(w[0]); x.c0 = fp_from_raw(w[1]); y.c1 = fp_from_raw(w[2]); y.c0 = fp_from_raw(w[3]);
    const Fp2C G3[6] = ZKV_FROB3;
    const Fp2 x3 = f2_mul(f2_conj(x), f2_const(G3[2])), y3 = f2_mul(f2_conj(y), f2_const(G3[3]));     // psi^3(B), affine
    const Fp2 k = f2_add(f2_mul(x, y), f2_one());
    uint32_t buf[48];
    MRef tm = m_ref(buf, 1);
    int out = 0;
    m_st_f2(tm, 0, f2_mul(x3, k)); m_st_f2(tm, 1, f2_neg(f2_mul(y3, k))); m_st_f2(tm, 2, k);
    if (miller_point_closes(tm, x, y)) out |= 1;
    m_st_f2(tm, 2, f2_zero());
    if (miller_point_closes(tm, x, y)) out |= 2;
    m_st_f2(tm, 0, f2_zero()); m_st_f2(tm, 1, y);
    if (miller_point_closes(tm, x, y)) out |= 4;
    m_st_f2(tm, 0, f2_mul(x3, k)); m_st_f2(tm, 1, f2_mul(y3, k)); m_st_f2(tm, 2, k);
    if (miller_point_closes(tm, x, y)) out |= 8;
    return out;
}
// 1/a through fp_inv (safegcd division steps) for a raw value given big-endian; loose != 0 feeds the second representation
// (a + p) of the Montgomery residue.  Returns 1 when the Fermat chain (fp_inv_fermat) gives the same canonical result.
int hs_fp_inv(const uint8_t* a32, int loose, uint8_t* out32) {
    uint32_t w[8]; load_be256(w, a32);
    Fp a = fp_from_raw(w);
    if (loose) {
        const uint32_t P[8] = ZKV_FP_P_LIMBS; const uint32_t P2[8] = ZKV_FP_2P_LIMBS;
        Fp t; uint32_t c = 0;
        for (int i = 0; i < 8; i++) t.v[i] = addc(a.v[i], P[i], c);
        if (!u256_geq(t.v, P2)) a = t;                  // only if the residue was below p
    }
    uint32_t r[8], r2[8];
    fp_to_raw(r, fp_inv(a)); fp_to_raw(r2, fp_inv_fermat(a));
    for (int i = 0; i < 8; i++) for (int k = 0; k < 4; k++) out32[31 - 4 * i - k] = r[i] >> (8 * k);
    int same = 1;
    for (int i = 0; i < 8; i++) same &= r[i] == r2[i];
    return same;
}
// fp_sqr against fp_mul(a, a) on ANY 256-bit operand (the multipliers accept lazy sums): 1 when the residues agree and stay below 2p
int hs_fp_sqr_check(const uint8_t* a32) {
    Fp a; load_be256(a.v, a32);
    const uint32_t P2[8] = ZKV_FP_2P_LIMBS;
    const Fp s = fp_sqr(a), m = fp_mul(a, a);
    return fp_eq(s, m) && !u256_geq(s.v, P2);
}
int hs_g2_in_subgroup(const uint8_t* q128) {     // EIP-197 order: x_im x_re y_im y_re; must be on twist
    uint32_t w[4][8];
    for (int k = 0; k < 4; k++) load_be256(w[k], q128 + 32 * k);
    Fp2 x, y; x.c1 = fp_from_raw(w[0]); x.c0 = fp_from_raw(w[1]); y.c1 = fp_from_raw(w[2]); y.c0 = fp_from_raw(w[3]);
    if (!g2_on_twist(x, y)) return -1;
    return g2_in_subgroup(x, y) ? 1 : 0;
}
}
