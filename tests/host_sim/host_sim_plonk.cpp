// Host build of the PLONK pre-pairing stage (stylus_zkvm_verifiers_amd/csrc/zkv_plonk.h) for CPU-side tests.  TEST ONLY: the
// shipped library never runs this on the host; it lets `-m "not gpu"` tests check the exact function k_plonk_prep executes
// (transcript, scalar algebra, MSMs) against oracle/plonk_model.py.
#define ZKV_COUNT_FP_MUL 1
#include <stdint.h>
#include <string.h>
#include "../../stylus_zkvm_verifiers_amd/csrc/zkv_host_vk.h"
#include "../../stylus_zkvm_verifiers_amd/csrc/zkv_plonk.h"

using namespace zkv;

static void wr_be(uint8_t* p, const uint32_t l[8]) { for (int i = 0; i < 8; i++) for (int k = 0; k < 4; k++) p[31 - 4 * i - k] = (uint8_t)(l[i] >> (8 * k)); }

static unsigned long long g_counts[3];
static PlonkKey* key_for(const uint8_t* vk, size_t vk_len) {
    if (vk_len < 7 * 32) return nullptr;
    uint32_t w7[7][8];
    for (int k = 0; k < 7; k++) host::be_to_limbs(w7[k], vk + 32 * k);
    const size_t n_c = w7[5][0];
    if (n_c > 1 || vk_len != 7 * 32 + (8 + n_c) * 64 + 256) return nullptr;
    static PlonkKeyRaw raw, last; static PlonkKey key; static bool have = false;
    memset(&raw, 0, sizeof raw);
    memcpy(raw.size, w7[0], 32); memcpy(raw.size_inv, w7[1], 32); memcpy(raw.gen, w7[2], 32); memcpy(raw.coset, w7[3], 32);
    raw.nb_public = w7[4][0]; raw.n_c = w7[5][0]; raw.cci = w7[6][0];
    for (size_t p = 0; p < 8 + n_c; p++) { host::be_to_limbs(raw.pts[p][0], vk + 224 + 64 * p); host::be_to_limbs(raw.pts[p][1], vk + 256 + 64 * p); }
    if (!have || memcmp(&raw, &last, sizeof raw) != 0) {     // the tables of a key (24 MB, 374 k chords) are built once per key
        plonk_setup_key(raw, key); plonk_setup_tables(key);
        last = raw; have = true;
    }
    return &key;
}
extern "C" {
// work of the last hsp_prepare call inside plonk_prepare: Fp multiplications, Fr multiplications, and the 32 x 32 + 64 multiply-adds
// (v_mad_u64_u32 on the device) both are made of -- the figures behind roofline.mulmod / roofline.achieved of bench.py --workload plonk_2p18
void hsp_prepare_counts(unsigned long long* out) { for (int i = 0; i < 3; i++) out[i] = g_counts[i]; }
// vk: the serialisation of include/zkv.h (7 words, 8 + n_c points, 256 bytes of G2); proof: 27 x 32 bytes; pub: 2 x 32 bytes.
// Returns -1 for a malformed key, 0 when the stage rejects, 1 when it produced the pairing inputs: out = D.x D.y Q.x Q.y (128 bytes,
// (0,0) = infinity).
int hsp_prepare(const uint8_t* vk, size_t vk_len, const uint8_t* proof, const uint8_t* pub, uint8_t* out128) {
    PlonkKey* kp = key_for(vk, vk_len);
    if (!kp) return -1;
    PlonkKey& key = *kp;
    uint32_t w[27][8], pb[2][8];
    for (int k = 0; k < 27; k++) host::be_to_limbs(w[k], proof + 32 * k);
    host::be_to_limbs(pb[0], pub); host::be_to_limbs(pb[1], pub + 32);
    PlonkOut o;
    const unsigned long long c0 = zkv_fp_mul_counter, d0 = zkv_mad_counter, e0 = zkv_fr_mul_counter;
    static uint32_t tabmem[PLONK_TAB_WORDS];
    const TabRef tab = {tabmem};
    const bool okp = plonk_prepare(key, w, pb, o, tab);
    g_counts[0] = zkv_fp_mul_counter - c0; g_counts[1] = zkv_fr_mul_counter - e0; g_counts[2] = zkv_mad_counter - d0;
    if (!okp) return 0;
    memset(out128, 0, 128);
    uint32_t r[8];
    G1A d, q; uint32_t d_inf, q_inf;
    g1j_to_affine(o.d, d, d_inf); g1j_to_affine(o.q, q, q_inf);
    if (!d_inf) { fp_to_raw(r, d.x); wr_be(out128, r); fp_to_raw(r, d.y); wr_be(out128 + 32, r); }
    if (!q_inf) { fp_to_raw(r, q.x); wr_be(out128 + 64, r); fp_to_raw(r, q.y); wr_be(out128 + 96, r); }
    return 1;
}
// Fr product through the device code (Montgomery in, Montgomery out) on canonical inputs: a * b mod r
void hsp_fr_mulmod(const uint8_t* a, const uint8_t* b, uint8_t* out) {
    uint32_t x[8], y[8], r[8];
    host::be_to_limbs(x, a); host::be_to_limbs(y, b);
    fr_to_raw(r, fr_mul(fr_from_raw_reduce(x), fr_from_raw_reduce(y)));
    wr_be(out, r);
}
// GLV split of a canonical scalar (zkv_scalar.h glv_split): out = |k1| (5 words), sign, |k2| (5 words), sign, little-endian words
void hsp_glv_split(const uint8_t* k32, uint32_t* out12) {
    uint32_t k[8], m1[5], m2[5], n1, n2;
    host::be_to_limbs(k, k32);
    glv_split(k, m1, n1, m2, n2);
    for (int i = 0; i < 5; i++) { out12[i] = m1[i]; out12[6 + i] = m2[i]; }
    out12[5] = n1; out12[11] = n2;
}
void hsp_fr_inv(const uint8_t* a, uint8_t* out) {
    uint32_t x[8], r[8];
    host::be_to_limbs(x, a);
    fr_to_raw(r, fr_inv(fr_from_raw_reduce(x)));
    wr_be(out, r);
}
// entry (a, b) of the joint table of key point p (PK_GEN = 9: the generator): a P + b phi(P) as 64 big-endian bytes; 1 if the point has a table
int hsp_joint_entry(const uint8_t* vk, size_t vk_len, int p, int a, int b, uint8_t* out64) {
    PlonkKey* kp = key_for(vk, vk_len);
    if (!kp || p < 0 || p > PK_POINTS || a < 0 || a > PK_JA || b < -PK_JA || b > PK_JA) return -1;
    if (kp->mult_inf[p]) return 0;
    const G1A e = kp->joint[p][a][b + PK_JA];
    uint32_t r[8];
    fp_to_raw(r, e.x); wr_be(out64, r); fp_to_raw(r, e.y); wr_be(out64 + 32, r);
    return 1;
}
}
