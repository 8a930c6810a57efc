// Host emulation of the 16-lanes-per-proof kernels (stylus_zkvm_verifiers_amd/csrc/k_wide.hip): the same zkv_tower_wide.h
// code the GPU runs, the twelve active lanes of a group played by twelve threads.  The DPP exchange inside a pair is a shared
// slot and a two-thread barrier; the lockstep of a wavefront (every load of a routine before its stores) and the work-group
// fence are a twelve-thread barrier (zkv_wide_host_barrier).  TEST ONLY.
#define ZKV_PAIRED 1
#include <atomic>
#include <stdint.h>
#include <string.h>
#include <thread>
#include <vector>
#include "../../stylus_zkvm_verifiers_amd/csrc/zkv_verify.h"

static constexpr int PAIRS = 6, MAX_SLICES = 4;
static thread_local uint32_t tl_par = 0, tl_pair = 0;          // tl_pair = slice * PAIRS + coefficient
static int g_threads = 2 * PAIRS;
struct Barrier {
    std::atomic<int> cnt{0}, gen{0};
    void wait(int n) {
        int g = gen.load(std::memory_order_acquire);
        if (cnt.fetch_add(1, std::memory_order_acq_rel) == n - 1) { cnt.store(0, std::memory_order_relaxed); gen.fetch_add(1, std::memory_order_acq_rel); }
        else while (gen.load(std::memory_order_acquire) == g) std::this_thread::yield();
    }
};
// two groups of threads: group 0 is the proof's (only, or consumer) wavefront, group 1 the producer wavefront of the two-wavefront Miller kernel
static Barrier g_pair_bar[2][PAIRS * MAX_SLICES], g_group_bar[2];
static volatile uint32_t g_xch[2][PAIRS * MAX_SLICES][2];
static uint32_t g_pair_tmp[PAIRS * MAX_SLICES][96];
static thread_local int tl_group = 0;
namespace zkv {
uint32_t zkv_parity() { return tl_par; }
uint32_t zkv_partner_u32(uint32_t x) {
    g_xch[tl_group][tl_pair][tl_par] = x; g_pair_bar[tl_group][tl_pair].wait(2);
    uint32_t r = g_xch[tl_group][tl_pair][tl_par ^ 1u]; g_pair_bar[tl_group][tl_pair].wait(2);
    return r;
}
void zkv_wide_host_barrier() { g_group_bar[tl_group].wait(g_threads); }
void zkv_wide_host_yield() { std::this_thread::yield(); }
uint32_t* zkv_wide_host_pair_tmp() { return g_pair_tmp[tl_pair]; }
}
#include "../../stylus_zkvm_verifiers_amd/csrc/zkv_tower_wide.h"
using namespace zkv;

struct Job { const VkTables* t; uint32_t flags; const uint32_t* norm48; const uint32_t* b32; int accept[2 * PAIRS * MAX_SLICES]; };
// group-shared memory: f, T, scratch, the slices' rows (LDS on the device) and the eight Fp12 slots of the final exponentiation (HBM on the device)
static uint32_t g_lds[96 + 48 + 13 * 16 + MAX_SLICES * 96 + 48], g_acc[96 + MAX_SLICES * 96 + 48], g_full[8 * 96];      // + 48: the slices' ninth limbs (W_RED_WORDS)

template <int S> static void lane(Job* j, uint32_t slice, uint32_t pair, uint32_t par) {
    tl_group = 0; tl_pair = slice * PAIRS + pair; tl_par = par;
    const int q = (int)pair;
    const WL w = {q, (int)slice};
    G1Norm n; Fp* nf[6] = {&n.axs, &n.ays, &n.lxs, &n.lys, &n.cxs, &n.cys};
    for (int k = 0; k < 6; k++) memcpy(nf[k]->v, j->norm48 + 8 * k, 32);
    Fp2 bx, by;
    memcpy(bx.h.v, j->b32 + 8 * par, 32); memcpy(by.h.v, j->b32 + 16 + 8 * par, 32);
    MRef fm = m_ref(g_lds + 8 * par, 1, 16), tm = m_ref(g_lds + 96 + 8 * par, 1, 16), sc = m_ref(g_lds + 144 + 8 * par, 1, 16);
    MRef red = m_ref(g_lds + 352 + 8 * par, 1, 16), red2 = m_ref(g_acc + 96 + 8 * par, 1, 16);
    miller_loop_w<S>(*j->t, j->flags, n, bx, by, fm, tm, sc, w, red);
    MRef ab = m_ref((uint32_t*)j->t->f_alpha_beta + 8 * par, 1, 16);
    MRef F = m_ref(g_full + 8 * par, 1, 16), E = m_ref(g_full + 96 + 8 * par, 1, 16), acc = m_ref(g_acc + 8 * par, 1, 16);
    w12_mul<S>(F, fm, ab, w, false, red);
    j->accept[2 * (slice * PAIRS + pair) + par] = final_exp_is_one_w<S>(F, E, m_off(E, 96), m_off(E, 192), m_off(E, 288), m_off(E, 384), acc, w, red2) ? 1 : 0;
}

template <int S> static int run_group(const void* tables, uint32_t flags, const uint32_t* norm48, const uint32_t* b32) {
    Job j; j.t = (const VkTables*)tables; j.flags = flags; j.norm48 = norm48; j.b32 = b32;
    g_threads = 2 * PAIRS * S;
    std::vector<std::thread> ts;
    for (uint32_t sl = 0; sl < (uint32_t)S; sl++) for (uint32_t p = 0; p < PAIRS; p++) for (uint32_t h = 0; h < 2; h++) if (sl || p || h) ts.emplace_back(lane<S>, &j, sl, p, h);
    lane<S>(&j, 0u, 0u, 0u);
    for (auto& t : ts) t.join();
    for (int k = 1; k < 2 * PAIRS * S; k++) if (j.accept[k] != j.accept[0]) return -1;
    return j.accept[0];
}
// The two-wavefront Miller kernel (k_miller_w64d): 48 consumer threads (group 0) and 48 producer threads (group 1) around the line
// table and the step counter, then the final exponentiation on the consumer's threads.
static uint32_t g_lines[ZKV_MILLER_STEPS * 48], g_tsc[48 + 13 * 16];
static volatile uint32_t g_ready;
static void dual_lane(Job* j, uint32_t group, uint32_t slice, uint32_t pair, uint32_t par) {
    tl_group = (int)group; tl_pair = slice * PAIRS + pair; tl_par = par;
    const WL w = {(int)pair, (int)slice};
    const bool do_ab = !(j->flags & (FL_A_INF | FL_B_INF));
    MRef lines = m_ref(g_lines + 8 * par, 1, 16);
    if (group == 1) {
        if (!do_ab) return;
        Fp2 bx, by;
        memcpy(bx.h.v, j->b32 + 8 * par, 32); memcpy(by.h.v, j->b32 + 16 + 8 * par, 32);
        MRef tm = m_ref(g_tsc + 8 * par, 1, 16), sc = m_ref(g_tsc + 48 + 8 * par, 1, 16);
        miller_lines_producer(bx, by, tm, sc, lines, &g_ready, (int)pair);
        return;
    }
    G1Norm n; Fp* nf[6] = {&n.axs, &n.ays, &n.lxs, &n.lys, &n.cxs, &n.cys};
    for (int k = 0; k < 6; k++) memcpy(nf[k]->v, j->norm48 + 8 * k, 32);
    MRef fm = m_ref(g_lds + 8 * par, 1, 16), sc = m_ref(g_lds + 96 + 8 * par, 1, 16), red = m_ref(g_lds + 352 + 8 * par, 1, 16), red2 = m_ref(g_acc + 96 + 8 * par, 1, 16);
    const bool fine = miller_loop_consumer<4>(j->t, j->flags, n, fm, sc, lines, &g_ready, w, red);
    MRef ab = m_ref((uint32_t*)j->t->f_alpha_beta + 8 * par, 1, 16);
    MRef F = m_ref(g_full + 8 * par, 1, 16), E = m_ref(g_full + 96 + 8 * par, 1, 16), acc = m_ref(g_acc + 8 * par, 1, 16);
    w12_mul<4>(F, fm, ab, w, false, red);
    const bool one = final_exp_is_one_w<4>(F, E, m_off(E, 96), m_off(E, 192), m_off(E, 288), m_off(E, 384), acc, w, red2);
    j->accept[2 * (slice * PAIRS + pair) + par] = fine ? (one ? 1 : 0) : -2;
}
extern "C" int hs3_pairing_w64d(const void* tables, uint32_t flags, const uint32_t* norm48, const uint32_t* b32) {
    Job j; j.t = (const VkTables*)tables; j.flags = flags; j.norm48 = norm48; j.b32 = b32;
    g_threads = 2 * PAIRS * 4;
    g_ready = 0;
    std::vector<std::thread> ts;
    for (uint32_t grp = 0; grp < 2; grp++)
        for (uint32_t sl = 0; sl < 4; sl++) for (uint32_t p = 0; p < PAIRS; p++) for (uint32_t h = 0; h < 2; h++) ts.emplace_back(dual_lane, &j, grp, sl, p, h);
    for (auto& t : ts) t.join();
    for (int k = 1; k < 2 * PAIRS * 4; k++) if (j.accept[k] != j.accept[0]) return -1;
    return j.accept[0];
}
// Miller loop + final exponentiation of one proof on the emulated 12-lane group (16 lanes per proof); -1 when the lanes disagree.
extern "C" int hs3_pairing(const void* tables, uint32_t flags, const uint32_t* norm48, const uint32_t* b32) { return run_group<1>(tables, flags, norm48, b32); }
// ... and on the emulated one-proof-per-wavefront group: four slices of six pairs, 48 threads
extern "C" int hs3_pairing_w64(const void* tables, uint32_t flags, const uint32_t* norm48, const uint32_t* b32) { return run_group<4>(tables, flags, norm48, b32); }
