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

static constexpr int PAIRS = 6;
static thread_local uint32_t tl_par = 0, tl_pair = 0;
struct Barrier {
    std::atomic<int> cnt{0}, gen{0};
    void wait(int n) {
        int g = gen.load(std::memory_order_acquire);
        if (cnt.fetch_add(1, std::memory_order_acq_rel) == n - 1) { cnt.store(0, std::memory_order_relaxed); gen.fetch_add(1, std::memory_order_acq_rel); }
        else while (gen.load(std::memory_order_acquire) == g) std::this_thread::yield();
    }
};
static Barrier g_pair_bar[PAIRS], g_group_bar;
static volatile uint32_t g_xch[PAIRS][2];
static uint32_t g_pair_tmp[PAIRS][96];
namespace zkv {
uint32_t zkv_parity() { return tl_par; }
uint32_t zkv_partner_u32(uint32_t x) {
    g_xch[tl_pair][tl_par] = x; g_pair_bar[tl_pair].wait(2);
    uint32_t r = g_xch[tl_pair][tl_par ^ 1u]; g_pair_bar[tl_pair].wait(2);
    return r;
}
void zkv_wide_host_barrier() { g_group_bar.wait(2 * PAIRS); }
uint32_t* zkv_wide_host_pair_tmp() { return g_pair_tmp[tl_pair]; }
}
#include "../../stylus_zkvm_verifiers_amd/csrc/zkv_tower_wide.h"
using namespace zkv;

struct Job { const VkTables* t; uint32_t flags; const uint32_t* norm48; const uint32_t* b32; int accept[2 * PAIRS]; };
// group-shared memory: f, T, scratch (LDS on the device) and the eight Fp12 slots of the final exponentiation (HBM on the device)
static uint32_t g_lds[96 + 48 + 13 * 16], g_acc[96], g_full[8 * 96];

static void lane(Job* j, uint32_t pair, uint32_t par) {
    tl_pair = pair; tl_par = par;
    const int q = (int)pair;
    G1Norm n; Fp* nf[6] = {&n.axs, &n.ays, &n.lxs, &n.lys, &n.cxs, &n.cys};
    for (int k = 0; k < 6; k++) memcpy(nf[k]->v, j->norm48 + 8 * k, 32);
    Fp2 bx, by;
    memcpy(bx.h.v, j->b32 + 8 * par, 32); memcpy(by.h.v, j->b32 + 16 + 8 * par, 32);
    MRef fm = m_ref(g_lds + 8 * par, 1, 16), tm = m_ref(g_lds + 96 + 8 * par, 1, 16), sc = m_ref(g_lds + 144 + 8 * par, 1, 16);
    miller_loop_w(*j->t, j->flags, n, bx, by, fm, tm, sc, q);
    MRef ab = m_ref((uint32_t*)j->t->f_alpha_beta + 8 * par, 1, 16);
    MRef F = m_ref(g_full + 8 * par, 1, 16), E = m_ref(g_full + 96 + 8 * par, 1, 16), acc = m_ref(g_acc + 8 * par, 1, 16);
    w12_mul(F, fm, ab, q, false);
    j->accept[2 * pair + par] = final_exp_is_one_w(F, E, m_off(E, 96), m_off(E, 192), m_off(E, 288), m_off(E, 384), acc, q) ? 1 : 0;
}

// Miller loop + final exponentiation of one proof on the emulated 12-lane group; -1 when the lanes disagree.
extern "C" int hs3_pairing(const void* tables, uint32_t flags, const uint32_t* norm48, const uint32_t* b32) {
    Job j; j.t = (const VkTables*)tables; j.flags = flags; j.norm48 = norm48; j.b32 = b32;
    std::vector<std::thread> ts;
    for (uint32_t p = 0; p < PAIRS; p++) for (uint32_t h = 0; h < 2; h++) if (p || h) ts.emplace_back(lane, &j, p, h);
    lane(&j, 0u, 0u);
    for (auto& t : ts) t.join();
    for (int k = 1; k < 2 * PAIRS; k++) if (j.accept[k] != j.accept[0]) return -1;
    return j.accept[0];
}
