// Host emulation of the lane-pair kernels (stylus_zkvm_verifiers_amd/csrc/k_pair.hip): the same ZKV_PAIRED code the
// GPU runs, with the two lanes of a pair played by two threads and the DPP operand exchange by a shared slot and a
// barrier.  TEST ONLY.
#define ZKV_PAIRED 1
#define ZKV_COUNT_FP_MUL 1
#include <atomic>
#include <stdint.h>
#include <string.h>
#include <thread>
#include "../../stylus_zkvm_verifiers_amd/csrc/zkv_verify.h"

static thread_local uint32_t tl_par = 0;
static volatile uint32_t g_xch[2];
static std::atomic<int> g_cnt{0}, g_gen{0};
static void pair_barrier() {
    int g = g_gen.load(std::memory_order_acquire);
    if (g_cnt.fetch_add(1, std::memory_order_acq_rel) == 1) { g_cnt.store(0, std::memory_order_relaxed); g_gen.fetch_add(1, std::memory_order_acq_rel); }
    else while (g_gen.load(std::memory_order_acquire) == g) std::this_thread::yield();
}
namespace zkv {
uint32_t zkv_parity() { return tl_par; }
uint32_t zkv_partner_u32(uint32_t x) {
    g_xch[tl_par] = x; pair_barrier();
    uint32_t r = g_xch[tl_par ^ 1u]; pair_barrier();
    return r;
}
}
using namespace zkv;

struct Job { const VkTables* t; uint32_t flags; const uint32_t* norm48; const uint32_t* b32; int sub_ok[2]; int accept[2]; unsigned long long muls[2][3], mads[2][3]; };

static void lane(Job* j, uint32_t par) {
    tl_par = par;
    G1Norm n; Fp* nf[6] = {&n.axs, &n.ays, &n.lxs, &n.lys, &n.cxs, &n.cys};
    for (int k = 0; k < 6; k++) memcpy(nf[k]->v, j->norm48 + 8 * k, 32);
    Fp2 bx, by;
    memcpy(bx.h.v, j->b32 + 8 * par, 32); memcpy(by.h.v, j->b32 + 16 + 8 * par, 32);
    unsigned long long c0 = zkv_fp_mul_counter, d0 = zkv_mad_counter;
    j->muls[par][0] = j->muls[par][1] = j->muls[par][2] = 0;
    j->mads[par][0] = j->mads[par][1] = j->mads[par][2] = 0;
    const bool sub_classic = (j->flags & FL_B_INF) ? true : g2_in_subgroup(bx, by);      // k_g2chk2 (kept for the 16-lane kernels)
    j->muls[par][0] = zkv_fp_mul_counter - c0; c0 = zkv_fp_mul_counter; j->mads[par][0] = zkv_mad_counter - d0; d0 = zkv_mad_counter;
    // lane-private half slots (f2w = 8) for f, T and the accumulator; full-layout slots (f2w = 16) for the cold values
    static thread_local uint32_t half[48 + 24];
    static uint32_t full[8 * 96];                    // shared by the two lanes like the HBM slots
    MRef fm = m_ref(half, 1, 8), tm = m_ref(half + 48, 1, 8);
    SoaRef norm = {j->norm48, 1, 0u}, bsrc = {j->b32 + 8 * par, 1, 0u};
    const bool sub = miller_loop_p(j->t, j->flags, norm, bsrc, fm, tm, true);        // the flat, fully inlined loop k_miller2 runs, with its subgroup verdict
    j->sub_ok[par] = sub == sub_classic ? (sub ? 1 : 0) : -1;                        // the two tests must agree
    if (!sub) { j->muls[par][1] = zkv_fp_mul_counter - c0; j->mads[par][1] = zkv_mad_counter - d0; j->accept[par] = 0; return; }
    MRef ab = m_ref((uint32_t*)j->t->f_alpha_beta + 8 * par, 1, 16);
    MRef F = m_ref(full + 8 * par, 1, 16);
    f12m_mul(F, fm, ab);
    j->muls[par][1] = zkv_fp_mul_counter - c0; c0 = zkv_fp_mul_counter; j->mads[par][1] = zkv_mad_counter - d0; d0 = zkv_mad_counter;
    static thread_local uint32_t acc9[54 * 64];     // the accumulator in resident 29-bit limbs, laid out as one lane's column of the LDS slot
    j->accept[par] = final_exp_prog_p(full, full + 96, 1, 4u * 8u * par, l9_ref(acc9)) ? 1 : 0;    // the interpreted program k_finalexp2 runs
    j->muls[par][2] = zkv_fp_mul_counter - c0; j->mads[par][2] = zkv_mad_counter - d0;
}
// Fp multiplications (a lane's Fp2 product counts 2, fp_mul 1) spent by BOTH lanes of the pair in g2chk, miller, finalexp
// during the last hs2_pairing call: the work unit of the secondary roofline (bench.py, roofline.mulmod).
static unsigned long long g_pair_muls[3], g_pair_mads[3];
extern "C" void hs2_stage_muls(unsigned long long* out) { for (int k = 0; k < 3; k++) out[k] = g_pair_muls[k]; }
// the 32 x 32 + 64 multiply-adds (v_mad_u64_u32) behind them, both lanes: the work unit of roofline.mulmod.issue_bound
extern "C" void hs2_stage_mads(unsigned long long* out) { for (int k = 0; k < 3; k++) out[k] = g_pair_mads[k]; }

extern "C" int hs2_pairing(const void* tables, uint32_t flags, const uint32_t* norm48, const uint32_t* b32, int* sub_ok) {
    Job j; j.t = (const VkTables*)tables; j.flags = flags; j.norm48 = norm48; j.b32 = b32;
    g_cnt = 0;
    std::thread t1(lane, &j, 1u);
    lane(&j, 0u);
    t1.join();
    for (int k = 0; k < 3; k++) { g_pair_muls[k] = j.muls[0][k] + j.muls[1][k]; g_pair_mads[k] = j.mads[0][k] + j.mads[1][k]; }
    if (j.sub_ok[0] != j.sub_ok[1] || j.sub_ok[0] < 0 || j.accept[0] != j.accept[1]) return -1;     // the pair, and the two subgroup tests, must agree
    *sub_ok = j.sub_ok[0];
    return j.accept[0];
}

// Lane-pair Fp2 arithmetic on raw operands at the edge of the multipliers' contract (any value below 4p: lazy sums of two
// loose values).  in: a0 a1 b0 b1 as 8 little-endian words each (Montgomery-domain integers, NOT reduced); out: 5 results x 2
// components x 8 words, brought out of Montgomery form (canonical): a*b, a^2 (needs a < 2p), xi*a (a < 2p), a+b and a-b (a, b < 2p).
struct EdgeJob { const uint32_t* in; uint32_t* out; int ok[2]; };
static void edge_lane(EdgeJob* j, uint32_t par) {
    tl_par = par;
    const uint32_t P2[8] = ZKV_FP_2P_LIMBS;
    Fp2 a, b;
    memcpy(a.h.v, j->in + 8 * par, 32); memcpy(b.h.v, j->in + 16 + 8 * par, 32);
    const bool reduced = !u256_geq(a.h.v, P2) && !u256_geq(b.h.v, P2);
    const bool both_reduced = (zkv_partner_u32(reduced ? 1u : 0u) != 0) && reduced;
    Fp2 r[5];
    r[0] = f2_mul(a, b);
    if (both_reduced) { r[1] = f2_sqr(a); r[2] = f2_mul_xi(a); r[3] = f2_add(a, b); r[4] = f2_sub(a, b); }
    else { r[1] = r[2] = r[3] = r[4] = f2_zero(); }
    int ok = 1;
    for (int k = 0; k < 5; k++) {
        if (u256_geq(r[k].h.v, P2)) ok = 0;                       // every result stays in the loose range
        fp_to_raw(j->out + 16 * k + 8 * par, r[k].h);
    }
    j->ok[par] = ok;
}
extern "C" int hs2_f2_edge(const uint32_t* in32, uint32_t* out80) {
    EdgeJob j; j.in = in32; j.out = out80;
    g_cnt = 0;
    std::thread t1(edge_lane, &j, 1u);
    edge_lane(&j, 0u);
    t1.join();
    return j.ok[0] && j.ok[1];
}

// Two or four proofs on one lane pair with a shared accumulator (miller_loop_pg, the aggregate check's Miller kernel) against the product of the
// two proofs' own Miller values from miller_loop_p with the fixed pairs switched off: 1 = equal (and both subgroup verdicts as given).
struct Job2 { const VkTables* t; int g; uint32_t mask, abmask; const uint32_t* norm96; const uint32_t* b96; int ok[2]; uint32_t fine[2]; };     // 48 words per proof in both arrays
static uint32_t g_tq[4 * 48];                   // the running points' rows: proof p at words [48 p, 48 p + 48), shared by the two lanes like HBM
static void lane2(Job2* j, uint32_t par) {
    tl_par = par;
    static thread_local uint32_t half[48 + 48 + 48 + 24];
    MRef fm = m_ref(half, 1, 8), f0 = m_ref(half + 48, 1, 8), prod = m_ref(half + 96, 1, 8), tm = m_ref(half + 144, 1, 8);
    // rows laid out with stride 1 and proof 1 exactly 48 words (192 bytes) after proof 0
    SoaRef norm = {j->norm96, 1, 0u}, bsrc = {j->b96 + 8 * par, 1, 0u};
    SoaRW tq = {g_tq, 1, 4u * 8u * par};
    const uint32_t fine = j->g == 4 ? miller_loop_pg<4>(j->mask, j->abmask, norm, bsrc, tq, 4u * 48u, fm) : miller_loop_pg<2>(j->mask, j->abmask, norm, bsrc, tq, 4u * 48u, fm);
    j->fine[par] = fine;
    f12m_set_one(prod);
    for (uint32_t p = 0; p < (uint32_t)j->g; p++) {
        if (!((j->mask >> p) & 1u)) continue;
        SoaRef n1 = {j->norm96 + 48 * p, 1, 0u}, b1 = {j->b96 + 48 * p + 8 * par, 1, 0u};
        const uint32_t fl = FL_ALIVE | FL_L_INF | FL_C_INF | (((j->abmask >> p) & 1u) ? 0u : (uint32_t)FL_A_INF);
        (void)miller_loop_p(j->t, fl, n1, b1, f0, tm, true);
        f12m_mul(prod, prod, f0);
    }
    bool same = true;
    for (int k = 0; k < 6; k++) same = f2_eq(m_ld_f2(fm, k), m_ld_f2(prod, k)) && same;
    j->ok[par] = same ? 1 : 0;
}
extern "C" int hs2_miller2(const void* tables, int g, uint32_t mask, uint32_t abmask, const uint32_t* norm96, const uint32_t* b96, uint32_t* fine) {
    Job2 j; j.t = (const VkTables*)tables; j.g = g; j.mask = mask; j.abmask = abmask; j.norm96 = norm96; j.b96 = b96;
    g_cnt = 0;
    std::thread t1(lane2, &j, 1u);
    lane2(&j, 0u);
    t1.join();
    if (j.ok[0] != j.ok[1] || j.fine[0] != j.fine[1]) return -1;
    *fine = j.fine[0];
    return j.ok[0];
}

// ---- resident 29-bit limbs (zkv_field.h "L9"): the one-pass linear combination and the limb product at the edge of their contracts.
// hs2_lincomb: n <= 6 terms, each nine signed 32-bit limbs (a normalised value or a lazy limb-wise sum / difference) with a small integer
// coefficient; c as at the call sites.  out9 = the nine limbs of the result.  One thread: the combination itself exchanges nothing.
template <int N> static void lincomb_n(const uint32_t* xs, const int32_t* ks, int c, uint32_t* out9) {
    LTerm t[N];
    for (int j = 0; j < N; j++) { t[j].x = xs + 9 * j; t[j].k = ks[j]; }
    const L9 r = l9_lincomb(t, c);
    for (int i = 0; i < 9; i++) out9[i] = r.l[i];
}
extern "C" int hs2_lincomb(int n, const uint32_t* xs, const int32_t* ks, int c, uint32_t* out9) {
    switch (n) {
        case 1: lincomb_n<1>(xs, ks, c, out9); return 1;
        case 2: lincomb_n<2>(xs, ks, c, out9); return 1;
        case 3: lincomb_n<3>(xs, ks, c, out9); return 1;
        case 4: lincomb_n<4>(xs, ks, c, out9); return 1;
        case 5: lincomb_n<5>(xs, ks, c, out9); return 1;
        case 6: lincomb_n<6>(xs, ks, c, out9); return 1;
    }
    return 0;
}
// hs2_l9_mul: the lane product l9_mul on a pair of threads.  in36: a0 a1 b0 b1 as nine limbs each (a: limbs below 2^30; b: normalised);
// out18: the two components' nine limbs.
struct L9Job { const uint32_t* in; uint32_t* out; };
static void l9_lane(L9Job* j, uint32_t par) {
    tl_par = par;
    L9 a, b;
    for (int i = 0; i < 9; i++) { a.l[i] = j->in[9 * par + i]; b.l[i] = j->in[18 + 9 * par + i]; }
    const L9 r = l9_mul(a, b);
    for (int i = 0; i < 9; i++) j->out[9 * par + i] = r.l[i];
}
extern "C" void hs2_l9_mul(const uint32_t* in36, uint32_t* out18) {
    L9Job j; j.in = in36; j.out = out18;
    g_cnt = 0;
    std::thread t1(l9_lane, &j, 1u);
    l9_lane(&j, 0u);
    t1.join();
}
