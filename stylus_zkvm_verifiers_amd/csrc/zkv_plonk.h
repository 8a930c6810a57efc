// SP1 PLONK path (BASELINE.json configs[4]): everything of a gnark-style BN254 PLONK verification that comes
// BEFORE the pairing -- proof parsing, scalar / point validation, the SHA-256 Fiat-Shamir transcript, the scalar-field
// algebra, the linearised-polynomial and KZG folding MSMs -- one proof per lane.  The result is the two G1 points of the final
// check  e(D, [1]_2) * e(-Q, [tau]_2) == 1,  which then take the ordinary lane-pair Miller loop and final exponentiation with the
// SRS's two G2 points as FIXED pairs (precomputed line tables: the same slots gamma and delta use in the Groth16 contexts).
//
// PARITY UNPINNED BY CONSTRUCTION: the reference has no PLONK code, key or proof (/root/reference/README.md:25,
// /root/reference/contracts/src/lib.rs:11); the entry-point shape is sp1/verifier.rs:16-29, 58-111, the algorithm gnark's published
// verifier (backend/plonk/bn254/verify.go, gnark-crypto kzg / fiat-shamir / hash_to_field), restated in oracle/plonk_model.py.
#pragma once
#include "zkv_verify.h"
#include "zkv_scalar.h"      // Fr, the GLV split

namespace zkv {

// ---------------------------------------------------------------- streaming SHA-256 for the transcripts (byte granular)
// One copy of the block function in the instruction stream: byte() and word_be() are inlined at some 170 places of plonk_prepare, and
// with the 64 unrolled rounds inlined into each of them the stage was 4 MB of straight-line code that every wavefront fetched once.
ZKV_HD_NI void sha_stream_block(uint32_t* h, uint32_t* w) { sha256_compress(h, w); for (int i = 0; i < 16; i++) w[i] = 0; }
struct ShaStream {
    uint32_t h[8], w[16]; uint32_t n;
    ZKV_HD void init() { sha256_init(h); n = 0; for (int i = 0; i < 16; i++) w[i] = 0; }
    ZKV_HD void byte(uint32_t b) {
        const uint32_t k = (n >> 2) & 15u;
        w[k] = (w[k] << 8) | (b & 255u);
        n++;
        if ((n & 63u) == 0) sha_stream_block(h, w);
    }
    ZKV_HD void word_be(uint32_t v) {                      // four bytes, most significant first
        if ((n & 3u) == 0) {
            w[(n >> 2) & 15u] = v; n += 4;
            if ((n & 63u) == 0) sha_stream_block(h, w);
        } else { byte(v >> 24); byte(v >> 16); byte(v >> 8); byte(v); }
    }
    ZKV_HD void limbs_be(const uint32_t* l) {               // a 256-bit value as 32 big-endian bytes
#pragma unroll 1
        for (int i = 7; i >= 0; i--) word_be(l[i]);
    }
    ZKV_HD void finish(uint32_t out[8]) {
        const uint64_t bits = (uint64_t)n * 8u;
        byte(0x80u);
        while ((n & 63u) != 56u) byte(0);
        word_be((uint32_t)(bits >> 32)); word_be((uint32_t)bits);
        for (int i = 0; i < 8; i++) out[i] = h[i];
    }
};
ZKV_HD void digest_to_limbs(const uint32_t h[8], uint32_t l[8]) { for (int i = 0; i < 8; i++) l[7 - i] = h[i]; }     // big-endian digest as an integer

// ---------------------------------------------------------------- verifying key on the device
constexpr int PK_S1 = 0, PK_S2 = 1, PK_S3 = 2, PK_QL = 3, PK_QR = 4, PK_QM = 5, PK_QO = 6, PK_QK = 7, PK_QCP = 8, PK_POINTS = 9;
struct PlonkKeyRaw {                // what zkv_sp1_plonk_ctx_create parsed out of the key bytes (canonical little-endian limbs)
    uint32_t size[8], size_inv[8], gen[8], coset[8];
    uint32_t nb_public, n_c, cci, pad;
    uint32_t pts[PK_POINTS][2][8];
};
struct PlonkKey {
    uint32_t size[8];               // domain size (exponent)
    uint32_t size_p2[8];            // size + 2
    Fr size_inv, gen, coset, gen_cci;       // gen_cci = generator^(nb_public + cci)
    uint32_t nb_public, n_c, valid, pad;
    G1A pts[PK_POINTS]; uint32_t inf[PK_POINTS];
    uint32_t raw[PK_POINTS][2][8];  // the same points as canonical integers, for the transcripts
    // Tables of the FIXED terms of the multi-scalar multiplications -- the nine key points and (last row) the G1 generator:
    G1A mult[PK_POINTS + 1][256];   // m * P, affine (m = 0 unused; 1 .. PK_JA feed the joint rows), built like a vk_x window row
    uint32_t mult_inf[PK_POINTS + 1];
    // a P + b phi(P), a = 0 .. 136, b = -136 .. 136 (index b + 136), affine; phi(P) = (beta x, y) = lambda P (GLV).  ONE addition serves
    // both GLV halves of a fixed term over TWO 4-bit windows (16 e' + e with signed digits |e| <= 8): 17 additions per term instead of
    // the 66 of a per-half walk (round 3's 9 x 17 table served one window: 33).  (0, 0) unused.  2.4 MB per point, 24 MB per context.
    G1A joint[PK_POINTS + 1][137][273];
};
constexpr int PK_JA = 136, PK_JB = 273;
constexpr int PK_GEN = PK_POINTS;
// A G1 point as the precompiles take it: coordinates < P, on the curve or (0,0) = infinity.  Returns false when invalid.
ZKV_HD bool plonk_g1(const uint32_t x[8], const uint32_t y[8], G1A& out, uint32_t& inf) {
    if (!raw_lt_p(x) || !raw_lt_p(y)) return false;
    if (raw_is_zero(x) && raw_is_zero(y)) { inf = 1; out.x = fp_zero(); out.y = fp_zero(); return true; }
    out.x = fp_from_raw(x); out.y = fp_from_raw(y); inf = 0;
    return g1_on_curve(out.x, out.y);
}
ZKV_HD void plonk_setup_key(const PlonkKeyRaw& r, PlonkKey& k) {
    bool ok = raw_lt_r(r.size_inv) && raw_lt_r(r.gen) && raw_lt_r(r.coset) && r.n_c <= 1 && r.nb_public == 2;
    for (int i = 0; i < 8; i++) { k.size[i] = r.size[i]; k.size_p2[i] = r.size[i]; }
    uint32_t c = 2;
    for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)k.size_p2[i] + c; k.size_p2[i] = (uint32_t)t; c = (uint32_t)(t >> 32); }
    k.size_inv = fr_from_raw(r.size_inv); k.gen = fr_from_raw(r.gen); k.coset = fr_from_raw(r.coset);
    uint32_t e[8] = {r.nb_public + r.cci, 0, 0, 0, 0, 0, 0, 0};
    k.gen_cci = fr_pow(k.gen, e, 32);
    k.nb_public = r.nb_public; k.n_c = r.n_c; k.pad = 0;
    for (int p = 0; p < PK_POINTS; p++) {
        for (int j = 0; j < 8; j++) { k.raw[p][0][j] = r.pts[p][0][j]; k.raw[p][1][j] = r.pts[p][1][j]; }
        if (p == PK_QCP && !r.n_c) { k.pts[p].x = fp_zero(); k.pts[p].y = fp_zero(); k.inf[p] = 1; continue; }
        ok = plonk_g1(r.pts[p][0], r.pts[p][1], k.pts[p], k.inf[p]) && ok;
    }
    k.valid = ok ? 1u : 0u;
}
// The multiples of point p (a key point, or the generator for p = PK_GEN) the joint rows are made of: independent per point, one lane
// each at set-up.  A key point has small order only if it is the point at infinity (the curve has prime order).
ZKV_HD void plonk_setup_mult(PlonkKey& k, int p) {
    G1A base; uint32_t binf = 0;
    if (p == PK_GEN) { base.x = fp_one(); Fp two = fp_zero(); two.v[0] = 2; base.y = fp_from_raw(two.v); }
    else { base = k.pts[p]; binf = k.inf[p]; }
    k.mult_inf[p] = (binf || !k.valid) ? 1u : 0u;
    if (k.mult_inf[p]) return;
    setup_window_row(base.x, base.y, 0, k.mult[p]);
}

// Row a of the joint table of point p (needs mult[p]): independent per (p, a), one lane each at set-up.  The 273 chords of a row share one
// field inversion (prefix products parked in the y slots about to be written).  a P = +-b phi(P) never happens: a -+ b lambda = 0 mod r
// has no solution with |a|, |b| <= 136 other than a = b = 0.
ZKV_HD void plonk_joint_row(PlonkKey& k, int p, int a) {
    G1A* row = k.joint[p][a];
    const Fp beta = ZKV_GLV_BETA;
    if (k.mult_inf[p]) return;                               // a term at infinity is skipped by the walk: its rows are never read
    const G1A A = k.mult[p][a];                              // a = 0: unused
    Fp prod = fp_one();
#pragma unroll 1
    for (int j = 0; j < PK_JB; j++) {
        const int b = j - PK_JA, m = b < 0 ? -b : b;
        if (!a || !b) continue;
        row[j].y = prod;
        prod = fp_mul(prod, fp_sub(fp_mul(k.mult[p][m].x, beta), A.x));
    }
    Fp inv = fp_inv(prod);
#pragma unroll 1
    for (int j = PK_JB - 1; j >= 0; j--) {
        const int b = j - PK_JA, m = b < 0 ? -b : b;
        if (!a && !b) { row[j].x = fp_zero(); row[j].y = fp_zero(); continue; }
        if (!b) { row[j] = A; continue; }
        const Fp bx = fp_mul(k.mult[p][m].x, beta), by = b < 0 ? fp_neg(k.mult[p][m].y) : k.mult[p][m].y;      // b phi(P)
        if (!a) { row[j].x = bx; row[j].y = by; continue; }
        const Fp den = fp_sub(bx, A.x);
        const Fp dinv = fp_mul(inv, row[j].y);
        inv = fp_mul(inv, den);
        const Fp lam = fp_mul(fp_sub(by, A.y), dinv);
        const Fp x3 = fp_sub(fp_sub(fp_sqr(lam), A.x), bx);
        row[j].x = x3;
        row[j].y = fp_sub(fp_mul(lam, fp_sub(A.x, x3)), A.y);
    }
}

ZKV_HD void plonk_setup_tables(PlonkKey& k) {               // everything in sequence (host builds; the device runs one lane per point / row)
#pragma unroll 1
    for (int p = 0; p <= PK_POINTS; p++) plonk_setup_mult(k, p);
#pragma unroll 1
    for (int p = 0; p <= PK_POINTS; p++)
#pragma unroll 1
        for (int a = 0; a <= PK_JA; a++) plonk_joint_row(k, p, a);
}

// ---------------------------------------------------------------- G1 helpers
// One term of a multi-scalar multiplication: an affine point (or infinity) and a canonical 256-bit scalar.
// fixed: the key's joint table [137][273] of the point, or null for a proof point, whose per-proof table sits in slot `slot` of the lane's
// table region -- to be built by this multiplication, or (`ready`) left there, affine, by an earlier one of the same proof.
struct MsmTerm { Fp x, y; uint32_t inf; uint32_t k[8]; const G1A* fixed; int slot; uint32_t ready; };

// 33 signed 4-bit digits of a magnitude below 2^131, packed 4 bits each as d + 8
ZKV_HD void glv_digits(const uint32_t (&m)[5], uint32_t (&dig)[5]) {
    uint32_t carry = 0;
#pragma unroll 1
    for (int w = 0; w < 5; w++) {
        uint32_t out = 0;
#pragma unroll 1
        for (int j = 0; j < 8; j++) {
            uint32_t d = ((m[w] >> (4 * j)) & 15u) + carry;          // 0 .. 16
            carry = d >= 8u ? 1u : 0u;
            out |= ((d + 8u) & 15u) << (4 * j);                        // d - 16 * carry + 8
        }
        dig[w] = out;
    }
}

// start + sum k_i P_i: Straus with SIGNED 4-BIT WINDOWS over the GLV halves of every scalar.  Per term the multiples P .. 8P are
// tabulated once (affine, see below) together with beta * x, which makes them the multiples of phi(P) = lambda P; each scalar splits
// into two halves below 2^128, so the shared doubling chain has 33 windows = 132 doublings instead of 256, and every window adds
// +-(|d| P) and +-(|d'| phi(P)) for every term -- every lane of the wavefront takes the same path (a one-bit-per-step loop executes
// each chord addition for all lanes although only half of them need it).
// The per-proof tables of the proof points live in ONE CONTIGUOUS REGION PER LANE (PLONK_TAB_WORDS words; global memory on the device,
// where the kernel gets it from the context): entry e of table `slot` is 24 consecutive words -- Jacobian (x, y, z) while the table
// is being built, then affine [x | y | beta x] -- so that a lookup with a lane-dependent digit reads 64 contiguous bytes.  Round 2 kept
// them in the kernel's private memory, which the hardware interleaves dword by dword across the 64 lanes of a wavefront: a lookup
// whose digit differs from lane to lane then touches 16 x 64 different cache lines, and the kernel moved 61 GB of HBM traffic per
// 2^18 proofs (profiles/round3_f_plonk_2p18_rocprofv3_pmc_summary.md) for 0.26 GB of input.
constexpr int PLONK_TAB_SLOTS = 5;                       // the most proof points any of the four multi-scalar multiplications has
constexpr int PLONK_TAB_WORDS = PLONK_TAB_SLOTS * 8 * 24;
constexpr int Z_SLOT = PLONK_TAB_SLOTS - 1;              // the slot the folding multiplication (slots 0..2) does not touch
struct TabRef { uint32_t* p; };
ZKV_HD Fp tab_ld(const TabRef& t, int slot, int e, int field) {
    const uint32_t* q = t.p + ((slot * 8 + e) * 3 + field) * 8;
    Fp r;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 a = ((const uint4*)q)[0], b = ((const uint4*)q)[1];       // 16-byte aligned: the region starts on a 256-byte boundary
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
#else
    for (int i = 0; i < 8; i++) r.v[i] = q[i];
#endif
    return r;
}
ZKV_HD void tab_st(const TabRef& t, int slot, int e, int field, const Fp& v) {
    uint32_t* q = t.p + ((slot * 8 + e) * 3 + field) * 8;
#if defined(__HIP_DEVICE_COMPILE__)
    ((uint4*)q)[0] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]); ((uint4*)q)[1] = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
#else
    for (int i = 0; i < 8; i++) q[i] = v.v[i];
#endif
}
ZKV_HD G1J tab_ld_j(const TabRef& t, int slot, int e) { G1J r; r.x = tab_ld(t, slot, e, 0); r.y = tab_ld(t, slot, e, 1); r.z = tab_ld(t, slot, e, 2); return r; }
ZKV_HD void tab_st_j(const TabRef& t, int slot, int e, const G1J& p) { tab_st(t, slot, e, 0, p.x); tab_st(t, slot, e, 1, p.y); tab_st(t, slot, e, 2, p.z); }
// multiples P .. 8P of one proof point, Jacobian
ZKV_HD void plonk_msm_table(const TabRef& tb, int slot, const MsmTerm& t) {
    G1J p; p.x = t.x; p.y = t.y; p.z = fp_one();
    const G1J m1 = g1j_dbl(p), m3 = g1j_dbl(m1), m2 = g1j_add_affine(m1, t.x, t.y);
    tab_st_j(tb, slot, 0, p); tab_st_j(tb, slot, 1, m1); tab_st_j(tb, slot, 2, m2); tab_st_j(tb, slot, 3, m3);
    tab_st_j(tb, slot, 4, g1j_add_affine(m3, t.x, t.y));
    const G1J m5 = g1j_dbl(m2);
    tab_st_j(tb, slot, 5, m5);
    tab_st_j(tb, slot, 6, g1j_add_affine(m5, t.x, t.y));
    tab_st_j(tb, slot, 7, g1j_dbl(m3));
}
// NV: how many tables this multiplication BUILDS at most (their slots are below NV): the prefix products of their normalisation are
// what lives in the lane's private memory.  Two proof points enter two multiplications each (Z, H_zeta_omega): the second use finds the
// table of the first (MsmTerm::ready) -- the callers place them in slots the multiplications in between leave alone.
template <int N, int NV> ZKV_HD G1J plonk_msm(const G1J& start, const MsmTerm (&t)[N], int n, const TabRef& tab) {
    static_assert(NV <= PLONK_TAB_SLOTS, "table region too small");
    uint32_t dig[N][2][5];                                // per half 33 signed digits, packed 4 bits each as d + 8 (0..15)
    uint32_t negs[N];                                     // bit 0 / 1: the first / second half is negative
    bool build = false;
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        if (t[i].inf) continue;
        if (!t[i].fixed && !t[i].ready) { plonk_msm_table(tab, t[i].slot, t[i]); build = true; }
        uint32_t m1[5], m2[5], n1, n2;
        glv_split(t[i].k, m1, n1, m2, n2);
        glv_digits(m1, dig[i][0]); glv_digits(m2, dig[i][1]);
        negs[i] = n1 | (n2 << 1);
    }
    // The per-proof tables become affine with ONE inversion for the whole multi-scalar multiplication (Montgomery's trick over the
    // seven Z of 2P..8P of every proof point: 7 multiplications per entry), so that all additions of a term are mixed additions
    // (11 instead of 16 multiplications).  Entries are never infinity: G1 has prime order.  z then holds beta * x.
    if (build) {
        const Fp beta = ZKV_GLV_BETA;
        Fp pre[NV][7];
        Fp run = fp_one();
#pragma unroll 1
        for (int i = 0; i < n; i++) {
            if (t[i].inf || t[i].fixed || t[i].ready) continue;
            const int s = t[i].slot;
#pragma unroll 1
            for (int m = 1; m < 8; m++) { pre[s][m - 1] = run; run = fp_mul(run, tab_ld(tab, s, m, 2)); }
        }
        Fp inv = fp_inv(run);
#pragma unroll 1
        for (int i = n - 1; i >= 0; i--) {
            if (t[i].inf || t[i].fixed || t[i].ready) continue;
            const int s = t[i].slot;
#pragma unroll 1
            for (int m = 7; m >= 1; m--) {
                const Fp zi = fp_mul(inv, pre[s][m - 1]);
                inv = fp_mul(inv, tab_ld(tab, s, m, 2));
                const Fp zi2 = fp_sqr(zi);
                const Fp ax = fp_mul(tab_ld(tab, s, m, 0), zi2);
                tab_st(tab, s, m, 0, ax);
                tab_st(tab, s, m, 1, fp_mul(tab_ld(tab, s, m, 1), fp_mul(zi2, zi)));
                tab_st(tab, s, m, 2, fp_mul(ax, beta));
            }
            tab_st(tab, s, 0, 2, fp_mul(tab_ld(tab, s, 0, 0), beta));
        }
    }
    G1J acc = g1j_infinity();
#pragma unroll 1
    for (int win = 32; win >= 0; win--) {
        acc = g1j_dbl(g1j_dbl(g1j_dbl(g1j_dbl(acc))));
#pragma unroll 1
        for (int i = 0; i < n; i++) {
            if (t[i].inf) continue;
            const int d1 = (int)((dig[i][0][win >> 3] >> (4 * (win & 7))) & 15u) - 8;      // -8 .. 7, magnitudes' digits
            const int d2 = (int)((dig[i][1][win >> 3] >> (4 * (win & 7))) & 15u) - 8;
            const int e1 = (negs[i] & 1u) ? -d1 : d1, e2 = (negs[i] & 2u) ? -d2 : d2;         // digits of k1 and k2
            if (t[i].fixed) {                                    // wave-uniform: the same term index in every lane
                if (win & 1) continue;                           // fixed terms take windows win + 1 and win together (the top one alone)
                int E1 = e1, E2 = e2;
                if (win < 32) {
                    const int w1 = win + 1;
                    const int h1 = (int)((dig[i][0][w1 >> 3] >> (4 * (w1 & 7))) & 15u) - 8, h2 = (int)((dig[i][1][w1 >> 3] >> (4 * (w1 & 7))) & 15u) - 8;
                    E1 += 16 * ((negs[i] & 1u) ? -h1 : h1); E2 += 16 * ((negs[i] & 2u) ? -h2 : h2);
                }
                if (E1 == 0 && E2 == 0) continue;
                const bool flip = E1 < 0;                        // -(|E1| P + (-E2) phi P)
                const G1A e = t[i].fixed[(flip ? -E1 : E1) * PK_JB + (flip ? -E2 : E2) + PK_JA];
                acc = g1j_add_affine(acc, e.x, flip ? fp_neg(e.y) : e.y);
                continue;
            }
            const int s = t[i].slot;
            if (e1) { const int m = (e1 < 0 ? -e1 : e1) - 1; const Fp y = tab_ld(tab, s, m, 1); acc = g1j_add_affine(acc, tab_ld(tab, s, m, 0), e1 < 0 ? fp_neg(y) : y); }
            if (e2) { const int m = (e2 < 0 ? -e2 : e2) - 1; const Fp y = tab_ld(tab, s, m, 1); acc = g1j_add_affine(acc, tab_ld(tab, s, m, 2), e2 < 0 ? fp_neg(y) : y); }
        }
    }
    return g1j_add(acc, start);
}
ZKV_HD void plonk_term(MsmTerm& t, const G1A& p, uint32_t inf, const Fr& k, int slot, uint32_t ready = 0) {
    t.x = p.x; t.y = p.y; t.inf = inf; t.fixed = nullptr; t.slot = slot; t.ready = ready; fr_to_raw(t.k, k);
}
ZKV_HD void plonk_key_term(MsmTerm& t, const PlonkKey& key, int p, const Fr& k) {      // a key point (or PK_GEN): table from the context
    t.x = key.mult[p][1].x; t.y = key.mult[p][1].y; t.inf = key.mult_inf[p]; t.fixed = &key.joint[p][0][0]; t.slot = -1; t.ready = 0; fr_to_raw(t.k, k);
}
// affine form + canonical coordinates for the transcripts
struct G1Bytes { uint32_t x[8], y[8]; };
ZKV_HD void plonk_affine(const G1J& p, G1A& a, uint32_t& inf, G1Bytes& raw) {
    g1j_to_affine(p, a, inf);
    if (inf) { for (int i = 0; i < 8; i++) { raw.x[i] = 0; raw.y[i] = 0; } return; }
    fp_to_raw(raw.x, a.x); fp_to_raw(raw.y, a.y);
}

// RFC 9380 expand_message_xmd(SHA-256, DST "BSB22-Plonk", 48 bytes) of a 64-byte point, reduced mod r
ZKV_HD Fr plonk_hash_to_field(const uint32_t x[8], const uint32_t y[8]) {
    const uint8_t DST[12] = {'B', 'S', 'B', '2', '2', '-', 'P', 'l', 'o', 'n', 'k', 11};
    uint32_t b0[8], b1[8], b2[8];
    ShaStream s;
    s.init();
    for (int i = 0; i < 16; i++) s.word_be(0);               // Z_pad: one block of zeros
    s.limbs_be(x); s.limbs_be(y);
    s.byte(0); s.byte(48); s.byte(0);                        // l_i_b_str = 48, then I2OSP(0, 1)
    for (int i = 0; i < 12; i++) s.byte(DST[i]);
    s.finish(b0);
    s.init();
    for (int i = 0; i < 8; i++) s.word_be(b0[i]);
    s.byte(1);
    for (int i = 0; i < 12; i++) s.byte(DST[i]);
    s.finish(b1);
    s.init();
    for (int i = 0; i < 8; i++) s.word_be(b0[i] ^ b1[i]);
    s.byte(2);
    for (int i = 0; i < 12; i++) s.byte(DST[i]);
    s.finish(b2);
    uint32_t hi[8], lo[8], sh[8] = {0, 0, 0, 0, 1, 0, 0, 0};             // 2^128
    digest_to_limbs(b1, hi);
    for (int i = 0; i < 4; i++) { lo[i] = b2[3 - i]; lo[4 + i] = 0; }   // the first 16 bytes of b2 as a 128-bit integer
    return fr_add(fr_mul(fr_from_raw_reduce(hi), fr_from_raw(sh)), fr_from_raw(lo));
}

// ---------------------------------------------------------------- the verifier up to the pairing
// words: the 27 proof words as canonical limbs.  pub: the two public inputs (program vkey unreduced, public-values hash).
// Returns false => VerificationFailed.  On success D and Q are the pairing's G1 inputs (Q already negated), Jacobian, Z = 0 for infinity:
// what the Miller loop wants of them -- x / y and 1 / y -- takes one inversion for both (k_plonk_prep), affine coordinates would take two more.
struct PlonkOut { G1J d, q; };
ZKV_HD_NI bool plonk_prepare(const PlonkKey& key, const uint32_t (&w)[27][8], const uint32_t (&pub)[2][8], PlonkOut& out, const TabRef& tab) {
    if (!key.valid) return false;
    if (!raw_lt_r(pub[0]) || !raw_lt_r(pub[1])) return false;
    const int SC[7] = {12, 13, 14, 15, 16, 19, 24};
    for (int i = 0; i < 7; i++) if (!raw_lt_r(w[SC[i]])) return false;
    // proof points: L R O H0 H1 H2 Z Hz Hzw BSB
    const int PT[10] = {0, 2, 4, 6, 8, 10, 17, 20, 22, 25};
    G1A pp[10]; uint32_t pinf[10];
    const int n_pts = key.n_c ? 10 : 9;
#pragma unroll 1
    for (int i = 0; i < n_pts; i++) if (!plonk_g1(w[PT[i]], w[PT[i] + 1], pp[i], pinf[i])) return false;
    const uint32_t n_c = key.n_c;
    // ---- challenges
    ShaStream s;
    uint32_t cg[8], cb[8], ca[8], cz[8], lim[8];
    s.init();
    s.byte('g'); s.byte('a'); s.byte('m'); s.byte('m'); s.byte('a');
#pragma unroll 1
    for (int p = 0; p < 8; p++) { s.limbs_be(key.raw[p][0]); s.limbs_be(key.raw[p][1]); }
    if (n_c) { s.limbs_be(key.raw[PK_QCP][0]); s.limbs_be(key.raw[PK_QCP][1]); }
    s.limbs_be(pub[0]); s.limbs_be(pub[1]);
#pragma unroll 1
    for (int i = 0; i < 6; i++) s.limbs_be(w[i]);
    s.finish(cg);
    s.init(); s.byte('b'); s.byte('e'); s.byte('t'); s.byte('a');
    for (int i = 0; i < 8; i++) s.word_be(cg[i]);
    s.finish(cb);
    s.init(); s.byte('a'); s.byte('l'); s.byte('p'); s.byte('h'); s.byte('a');
    for (int i = 0; i < 8; i++) s.word_be(cb[i]);
    if (n_c) { s.limbs_be(w[25]); s.limbs_be(w[26]); }
    s.limbs_be(w[17]); s.limbs_be(w[18]);
    s.finish(ca);
    s.init(); s.byte('z'); s.byte('e'); s.byte('t'); s.byte('a');
    for (int i = 0; i < 8; i++) s.word_be(ca[i]);
#pragma unroll 1
    for (int i = 6; i < 12; i++) s.limbs_be(w[i]);
    s.finish(cz);
    digest_to_limbs(cg, lim); const Fr gamma = fr_from_raw_reduce(lim);
    digest_to_limbs(cb, lim); const Fr beta = fr_from_raw_reduce(lim);
    digest_to_limbs(ca, lim); const Fr alpha = fr_from_raw_reduce(lim);
    digest_to_limbs(cz, lim); const Fr zeta = fr_from_raw_reduce(lim);
    // ---- public-input polynomial at zeta
    const Fr one = fr_one();
    const Fr zeta_n = fr_pow(zeta, key.size, 64);
    const Fr zh = fr_sub(zeta_n, one);
    Fr den = fr_sub(zeta, one);
    if (fr_is_zero(den)) return false;
    // the three denominators zeta - 1, zeta - w, zeta - w^(nb_public + cci) share one inversion
    const Fr d1 = fr_sub(zeta, key.gen), d2 = fr_sub(zeta, key.gen_cci);
    if (fr_is_zero(d1) || (n_c && fr_is_zero(d2))) return false;
    const Fr d2e = n_c ? d2 : one;
    const Fr p01 = fr_mul(den, d1);
    const Fr inv_all = fr_inv(fr_mul(p01, d2e));
    const Fr i2 = fr_mul(inv_all, p01);                    // 1 / d2
    const Fr i01 = fr_mul(inv_all, d2e);                   // 1 / (den d1)
    const Fr i0 = fr_mul(i01, d1), i1 = fr_mul(i01, den);
    const Fr zn = fr_mul(zh, key.size_inv);                // (zeta^n - 1) / n
    const Fr lag0 = fr_mul(zn, i0);
    Fr pi = fr_mul(lag0, fr_from_raw(pub[0]));
    pi = fr_add(pi, fr_mul(fr_mul(fr_mul(zn, i1), key.gen), fr_from_raw(pub[1])));
    if (n_c) pi = fr_add(pi, fr_mul(fr_mul(fr_mul(zn, i2), key.gen_cci), plonk_hash_to_field(w[25], w[26])));
    const Fr l = fr_from_raw(w[12]), r = fr_from_raw(w[13]), o = fr_from_raw(w[14]), s1 = fr_from_raw(w[15]), s2 = fr_from_raw(w[16]);
    const Fr zu = fr_from_raw(w[19]), qcpz = fr_from_raw(w[24]);
    // ---- opening of the linearised polynomial and the scalars of its digest
    const Fr a2l0 = fr_mul(fr_mul(lag0, alpha), alpha);
    const Fr t1 = fr_add(fr_add(fr_mul(beta, s1), l), gamma), t2 = fr_add(fr_add(fr_mul(beta, s2), r), gamma);
    const Fr at = fr_mul(fr_mul(alpha, t1), t2);
    const Fr lin_eval = fr_neg(fr_sub(fr_add(fr_mul(fr_mul(at, fr_add(o, gamma)), zu), pi), a2l0));
    const Fr _s1 = fr_mul(fr_mul(at, beta), zu);
    const Fr bz = fr_mul(beta, zeta), bzu = fr_mul(bz, key.coset), bzu2 = fr_mul(bzu, key.coset);
    const Fr _s2 = fr_neg(fr_mul(fr_mul(fr_mul(alpha, fr_add(fr_add(bz, l), gamma)), fr_add(fr_add(bzu, r), gamma)), fr_add(fr_add(bzu2, o), gamma)));
    const Fr coeff_z = fr_add(a2l0, _s2);
    const Fr zn2 = fr_pow(zeta, key.size_p2, 64);
    const Fr k0 = fr_neg(zh), k1 = fr_mul(zn2, k0), k2 = fr_mul(zn2, k1);
    // ---- linearised polynomial digest: qcp Pi2 + l Ql + r Qr + lr Qm + o Qo + Qk + _s1 S3 + coeff_z Z + k0 H0 + k1 H1 + k2 H2
    G1J qk = g1j_infinity();
    if (!key.inf[PK_QK]) { qk.x = key.pts[PK_QK].x; qk.y = key.pts[PK_QK].y; qk.z = fp_one(); }
    G1A lin_a, fold_a; uint32_t lin_inf, fold_inf; G1Bytes lin_b, fold_b;
    {
        MsmTerm t[10];
        plonk_key_term(t[0], key, PK_QL, l); plonk_key_term(t[1], key, PK_QR, r);
        plonk_key_term(t[2], key, PK_QM, fr_mul(l, r)); plonk_key_term(t[3], key, PK_QO, o);
        plonk_key_term(t[4], key, PK_S3, _s1); plonk_term(t[5], pp[6], pinf[6], coeff_z, Z_SLOT);     // Z's table stays for the last multiplication but one
        plonk_term(t[6], pp[3], pinf[3], k0, 0); plonk_term(t[7], pp[4], pinf[4], k1, 1); plonk_term(t[8], pp[5], pinf[5], k2, 2);
        if (n_c) plonk_term(t[9], pp[9], pinf[9], qcpz, 3);
        plonk_affine((plonk_msm<10, 5>(qk, t, n_c ? 10 : 9, tab)), lin_a, lin_inf, lin_b);
    }
    // ---- fold the openings at zeta: gamma_kzg = H("gamma" || zeta || digests || values || zu)
    uint32_t zr[8], ch[8];
    s.init(); s.byte('g'); s.byte('a'); s.byte('m'); s.byte('m'); s.byte('a');
    fr_to_raw(zr, zeta); s.limbs_be(zr);
    s.limbs_be(lin_b.x); s.limbs_be(lin_b.y);
#pragma unroll 1
    for (int i = 0; i < 6; i++) s.limbs_be(w[i]);
    s.limbs_be(key.raw[PK_S1][0]); s.limbs_be(key.raw[PK_S1][1]); s.limbs_be(key.raw[PK_S2][0]); s.limbs_be(key.raw[PK_S2][1]);
    if (n_c) { s.limbs_be(key.raw[PK_QCP][0]); s.limbs_be(key.raw[PK_QCP][1]); }
    uint32_t lr8[8]; fr_to_raw(lr8, lin_eval); s.limbs_be(lr8);
    s.limbs_be(w[12]); s.limbs_be(w[13]); s.limbs_be(w[14]); s.limbs_be(w[15]); s.limbs_be(w[16]);
    if (n_c) s.limbs_be(w[24]);
    s.limbs_be(w[19]);
    s.finish(ch);
    digest_to_limbs(ch, lim); const Fr g = fr_from_raw_reduce(lim);
    Fr folded_eval;
    {
        const Fr g2 = fr_mul(g, g), g3 = fr_mul(g2, g), g4 = fr_mul(g3, g), g5 = fr_mul(g4, g), g6 = fr_mul(g5, g);
        folded_eval = fr_add(lin_eval, fr_add(fr_add(fr_mul(g, l), fr_mul(g2, r)), fr_add(fr_mul(g3, o), fr_add(fr_mul(g4, s1), fr_mul(g5, s2)))));
        if (n_c) folded_eval = fr_add(folded_eval, fr_mul(g6, qcpz));
        MsmTerm t[6];
        plonk_term(t[0], pp[0], pinf[0], g, 0); plonk_term(t[1], pp[1], pinf[1], g2, 1); plonk_term(t[2], pp[2], pinf[2], g3, 2);
        plonk_key_term(t[3], key, PK_S1, g4); plonk_key_term(t[4], key, PK_S2, g5);
        if (n_c) plonk_key_term(t[5], key, PK_QCP, g6);
        G1J linj = g1j_infinity();
        if (!lin_inf) { linj.x = lin_a.x; linj.y = lin_a.y; linj.z = fp_one(); }
        plonk_affine((plonk_msm<6, 3>(linj, t, n_c ? 6 : 5, tab)), fold_a, fold_inf, fold_b);
    }
    // ---- batch the two openings: lambda = H(folded digest || H_zeta || Z || H_zeta_omega || zeta || gamma_kzg) mod r
    s.init();
    s.limbs_be(fold_b.x); s.limbs_be(fold_b.y);
    s.limbs_be(w[20]); s.limbs_be(w[21]); s.limbs_be(w[17]); s.limbs_be(w[18]); s.limbs_be(w[22]); s.limbs_be(w[23]);
    s.limbs_be(zr);
    uint32_t gr[8]; fr_to_raw(gr, g); s.limbs_be(gr);
    s.finish(ch);
    digest_to_limbs(ch, lim); const Fr lam = fr_from_raw_reduce(lim);
    const Fr evals = fr_add(folded_eval, fr_mul(lam, zu));
    G1J dj, qj;
    {
        MsmTerm t[4];
        plonk_term(t[0], pp[6], pinf[6], lam, Z_SLOT, 1);                    // built by the first multiplication
        plonk_key_term(t[1], key, PK_GEN, fr_neg(evals));
        plonk_term(t[2], pp[7], pinf[7], zeta, 0);
        plonk_term(t[3], pp[8], pinf[8], fr_mul(lam, fr_mul(zeta, key.gen)), 1);
        G1J fj = g1j_infinity();
        if (!fold_inf) { fj.x = fold_a.x; fj.y = fold_a.y; fj.z = fp_one(); }
        dj = plonk_msm<4, 3>(fj, t, 4, tab);
        MsmTerm u1[1];
        plonk_term(u1[0], pp[8], pinf[8], lam, 1, 1);                        // H_zeta_omega's table: just built
        G1J hz = g1j_infinity();
        if (!pinf[7]) { hz.x = pp[7].x; hz.y = pp[7].y; hz.z = fp_one(); }
        qj = plonk_msm<1, 1>(hz, u1, 1, tab);
        qj.y = fp_neg(qj.y);
    }
    out.d = dj; out.q = qj;
    return true;
}

}  // namespace zkv
