// Host build of the aggregate-check arithmetic the GPU runs (stylus_zkvm_verifiers_amd/csrc/zkv_agg.h): coefficient derivation,
// the GLV scalar multiplication, the lanes' shares of E and the three-point normalisation.  TEST ONLY.
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include "../../stylus_zkvm_verifiers_amd/csrc/zkv_host_vk.h"
#include "../../stylus_zkvm_verifiers_amd/csrc/zkv_agg.h"
using namespace zkv;

static void be_to_limbs(uint32_t l[8], const uint8_t* p) { load_be256(l, p); }
static void fp_to_be(uint8_t* o, const Fp& a) {
    uint32_t r[8];
    fp_to_raw(r, a);
    for (int k = 0; k < 8; k++) { uint32_t v = r[7 - k]; o[4 * k] = v >> 24; o[4 * k + 1] = v >> 16; o[4 * k + 2] = v >> 8; o[4 * k + 3] = v; }
}
static int affine_out(const G1J& p, uint8_t* out64) {
    G1A a; uint32_t inf;
    g1j_to_affine(p, a, inf);
    fp_to_be(out64, a.x); fp_to_be(out64 + 32, a.y);
    return inf ? 1 : 0;
}

extern "C" void hsa_coeff(const uint8_t* seed32, uint32_t call, uint32_t index, uint64_t* r) {
    AggSeed s;
    for (int i = 0; i < 8; i++) s.w[i] = load_be32(seed32 + 4 * i);
    s.call = call;
    agg_coeff(s, index, r[0], r[1]);
}
// r1 P + r2 phi(P); returns 1 for infinity
extern "C" int hsa_mul(const uint8_t* xy64, uint64_t r1, uint64_t r2, uint8_t* out64) {
    uint32_t x[8], y[8];
    be_to_limbs(x, xy64); be_to_limbs(y, xy64 + 32);
    return affine_out(agg_mul(fp_from_raw(x), fp_from_raw(y), r1, r2), out64);
}
// the lanes' shares summed (sub = 16, 32 or 64 lanes per sub-batch): E = (S1 - c) alpha + S2 phi(alpha)
extern "C" int hsa_e(const uint8_t* alpha64, uint32_t sub, uint64_t s1lo, uint32_t s1hi, uint64_t s2lo, uint32_t s2hi, uint32_t c, uint8_t* out64) {
    static VkRaw vk; static AggTables t;
    memset(&vk, 0, sizeof vk);
    be_to_limbs(vk.alpha[0], alpha64); be_to_limbs(vk.alpha[1], alpha64 + 32);
    for (int j = 0; j < AGG_ALPHA_POW; j++) setup_agg_alpha(vk, t, j);
    G1J acc = g1j_infinity();
    for (uint32_t lane = 0; lane < sub; lane++) acc = g1j_add(acc, agg_e_share(t, lane, sub, s1lo, s1hi, s2lo, s2hi, c));
    return affine_out(acc, out64);
}
// (x/y, 1/y) of three points given as affine (or all-zero = infinity) after a random Jacobian rescaling; out: 6 x 32 bytes + flags
extern "C" uint32_t hsa_norm3(const uint8_t* pts192, const uint8_t* z96, uint8_t* out192) {
    G1J p[3];
    for (int k = 0; k < 3; k++) {
        uint32_t x[8], y[8], z[8];
        be_to_limbs(x, pts192 + 64 * k); be_to_limbs(y, pts192 + 64 * k + 32); be_to_limbs(z, z96 + 32 * k);
        if (raw_is_zero(x) && raw_is_zero(y)) { p[k] = g1j_infinity(); continue; }
        const Fp Z = fp_from_raw(z), Z2 = fp_sqr(Z);
        p[k].x = fp_mul(fp_from_raw(x), Z2); p[k].y = fp_mul(fp_from_raw(y), fp_mul(Z2, Z)); p[k].z = Z;
    }
    uint32_t flags = FL_ALIVE;
    G1Norm o;
    agg_normalize3(p[0], p[1], p[2], flags, o);
    const Fp* f[6] = {&o.axs, &o.ays, &o.lxs, &o.lys, &o.cxs, &o.cys};
    for (int k = 0; k < 6; k++) fp_to_be(out192 + 32 * k, *f[k]);
    return flags;
}

// r = r1 + r2 lambda mod r as a canonical big-endian integer
extern "C" void hsa_coeff_fr(uint64_t r1, uint64_t r2, uint8_t* out32) {
    uint32_t l[8];
    fr_to_raw(l, agg_coeff_fr(r1, r2));
    for (int k = 0; k < 8; k++) { uint32_t v = l[7 - k]; out32[4 * k] = v >> 24; out32[4 * k + 1] = v >> 16; out32[4 * k + 2] = v >> 8; out32[4 * k + 3] = v; }
}
// the lanes' shares of U = R base + T_0 IC_a + T_1 IC_b summed, on the RISC Zero key (vm 0: control root / id given) or the SP1 key (vm 1)
extern "C" int hsa_u(int vm, const uint8_t* cr, const uint8_t* cid, uint32_t sub, const uint8_t* R32, const uint8_t* T64, uint8_t* out64) {
    static VkTables* vt = (VkTables*)malloc(sizeof(VkTables));
    static AggTables* at = (AggTables*)malloc(sizeof(AggTables));
    static int built = -1;
    if (built != vm) {
        VkRaw raw;
        if (vm == 0) { uint8_t lo[16], hi[16]; host::split_digest(cr, lo, hi); host::fill_vk_risc0(raw, lo, hi, cid); }
        else host::fill_vk_sp1(raw);
        memset(vt, 0, sizeof *vt); memset(at, 0, sizeof *at);
        setup_base(raw, *vt);
        for (uint32_t b = 0; b < raw.n_var; b++) for (int w = 0; w < MSM_MAX_WINDOWS; w++) setup_msm_row(raw, *vt, (int)b, w);
        for (int w = 0; w < MSM_MAX_WINDOWS; w++) setup_agg_base_row(*vt, *at, w);
        built = vm;
    }
    uint32_t R[8], T[2][8];
    be_to_limbs(R, R32); be_to_limbs(T[0], T64); be_to_limbs(T[1], T64 + 32);
    G1J acc = g1j_infinity();
    for (uint32_t lane = 0; lane < sub; lane++) acc = g1j_add(acc, agg_u_share(*vt, *at, lane, sub, R, T));
    return affine_out(acc, out64);
}

// k P through the GLV walk of the ecMul kernel (g1_mul_glv) for any 256-bit k
extern "C" int hsa_ecmul_glv(const uint8_t* xy64, const uint8_t* k32, uint8_t* out64) {
    uint32_t x[8], y[8], k[8];
    be_to_limbs(x, xy64); be_to_limbs(y, xy64 + 32); be_to_limbs(k, k32);
    return affine_out(g1_mul_glv(fp_from_raw(x), fp_from_raw(y), k), out64);
}
