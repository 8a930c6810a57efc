// C ABI of libzkv_mi355x.so (include/zkv.h): verifier contexts, batch entry points, chunked stage pipeline.
// Host code here only marshals bytes, runs the once-per-context SHA-256 chain (initialize) and enqueues
// kernels; every field/curve/pairing operation runs in the HIP kernels.  No CPU fallback exists.
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/zkv.h"
#include "zkv_host_abi.h"
#include "zkv_host_vk.h"
#include "zkv_internal.h"
#include "zkv_plonk.h"
#include "zkv_agg.h"
#include <sys/random.h>

using namespace zkv;

#define ZKV_EXPORT extern "C" __attribute__((visibility("default")))

struct zkv_ctx {
    int vm = 0, device = 0, lanes = 0;       // lanes: 0 = library default, 1 or 2 = lanes per proof for the Fp2-heavy stages
    bool initialized = false, id_ge_r = false;
    uint8_t control_root_0[16] = {0}, control_root_1[16] = {0}, control_id[32] = {0}, selector[4] = {0};
    Risc0Consts consts;
    // ZKV_VM_RISC0_SET: n_inst verifier instances sharing the VK tables and the workspace
    std::vector<InstRaw> inst_raw;
    std::vector<InstTab> inst_host;          // copy of the device table (selectors derived on the device) for the getters
    InstTab* d_inst = nullptr;
    uint32_t* d_inst_idx = nullptr;
    uint8_t gvk[448 + 64 * MAX_IC] = {0};    // ZKV_VM_GROTH16: the caller's verification key
    uint32_t g_n_ic = 0; bool g_negate = false, vk_invalid = false;
    // device side (created lazily on the first compute call)
    bool dev_ready = false;
    hipStream_t stream = nullptr;
    VkTables* d_tab = nullptr;
    G1A* d_msm16 = nullptr;                                  // the vk_x stage's 16-bit window rows (Msm16; null: 8-bit walk)
    Msm16 m16 = {nullptr, {0, 0, 0, 0, 0}};
    Workspace ws = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    hipStream_t side = nullptr;                              // small chunks: the G2 subgroup check runs beside the MSM
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    uint8_t *d_blob = nullptr, *d_a = nullptr, *d_b = nullptr, *d_pv = nullptr, *d_status = nullptr, *d_recv = nullptr;
    uint64_t *d_off = nullptr, *d_pvoff = nullptr;
    size_t blob_cap = 0, pv_cap = 0;
    // wire layer (eth_call batches): calldata blob, its offsets, decoded lengths / methods
    uint8_t *d_cd[2] = {nullptr, nullptr}, *d_kind = nullptr, *d_st_all = nullptr, *d_rv_all = nullptr;
    uint64_t* d_cdoff[2] = {nullptr, nullptr};
    uint32_t *d_len = nullptr, *d_pvlen = nullptr;
    size_t cd_cap[2] = {0, 0}, st_all_cap = 0, rv_all_cap = 0;
    hipStream_t copy_stream = nullptr;                       // H2D of calldata chunk k+1 overlaps the kernels of chunk k
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_decoded[2] = {nullptr, nullptr};
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, ev_wire[2] = {nullptr, nullptr};
    bool wire_timed = false;
    // The workspace is shared by every call on this context, and the *_dev entry points run on caller-chosen streams:
    // each call first makes its stream wait for the previous call's last kernel (ev_done), then records ev_done again.
    hipEvent_t ev_done = nullptr;
    bool has_done = false;
    // host-buffer batches (run_host_batch): whole-batch staging in HBM, filled segment by segment on copy_stream while the previous
    // segment is verified
    uint8_t* hb[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};     // seals, seal offsets, in_a, in_b, public values, pv offsets
    size_t hb_cap[6] = {0, 0, 0, 0, 0, 0};
    hipEvent_t ev_seg[2] = {nullptr, nullptr};
    // ZKV_VM_SP1_PLONK: parsed verifying key, the SRS's two G2 points (reference word order) and the verifier hash
    PlonkKeyRaw pk_raw; uint8_t pk_g2[256] = {0}, plonk_hash[32] = {0};
    PlonkKey* d_pkey = nullptr;
    uint32_t* d_plonk_tab = nullptr;                           // per-proof window tables of the PLONK stage (PLONK_TAB_WORDS words per proof in flight)
    // Aggregate check (zkv_agg.h, zkv_ctx_set_aggregate_check): key tables, per-proof rows, the pseudo-proofs' workspace (one per
    // sub-batch), their statuses and the counters {sub-batches checked, sub-batches failed}
    bool agg_on = false, agg_key_ok = false;
    uint32_t agg_sub = 32;                                     // proofs per sub-batch: 16, 32, 64, 128 or 256
    bool agg_auto = false;                                     // enable = 1: the size follows the failure rate seen so far (agg_adapt)
    unsigned long long agg_seen[2] = {0, 0};                   // counters at the last adaptation
    bool agg_resnap = false;                                   // just switched on: the next look only takes the counters as they are
    bool agg_look = false;                                     // one look at the counters per CALL (order_after_previous arms it): a later chunk of the same call
                                                               // could find the previous chunk's k_agg_mark half-way through its two counters
    uint32_t agg_pause = 0, agg_pause_len = 0;                 // automatic mode: chunks still to run WITHOUT the check (too many sub-batches fail), and the length of that pause
    bool agg_os_seed = false;                                  // the secret came from the operating system: it is drawn afresh every AGG_REKEY_CHUNKS chunks
    uint32_t agg_key_age = 0;
    AggTables* d_agg_tab = nullptr;
    uint32_t* d_agg = nullptr;
    Workspace ws2 = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    uint8_t* d_status2 = nullptr;
    // proofs of failed sub-batches, gathered into a dense workspace for the ordinary kernels: own PREP rows, flags and statuses; the
    // scratch rows (norm, f, fe) are the chunk workspace's
    Workspace ws3 = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    uint8_t* d_status3 = nullptr;
    uint32_t* d_agg_idx = nullptr;
    unsigned long long* d_agg_cnt = nullptr;
    size_t agg_cap = 0;
    AggSeed agg_seed = {{0, 0, 0, 0, 0, 0, 0, 0}, 0};
    // ZKV_VM_MIXED: one RISC Zero and one SP1 verifier behind a per-proof VM tag; mx[] are the demultiplexing buffers
    // Sharded (multi-device) context: `shards` single-device contexts of one verifier behind the ordinary batch entry points
    // (zkv_ctx_create_sharded).  sh[] is the per-shard state of device-resident batches: staging rows on the shard's device, a copy
    // stream (the transfer of piece k + 1 runs behind the kernels of piece k), the shard's compute stream and its events.
    struct ShardDev {
        uint8_t* row[4] = {nullptr, nullptr, nullptr, nullptr}; size_t row_cap[4] = {0, 0, 0, 0};
        uint8_t *st = nullptr, *rv = nullptr; size_t st_cap = 0, rv_cap = 0;
        hipStream_t copy = nullptr, run = nullptr;
        hipEvent_t ev_piece[2] = {nullptr, nullptr}, ev_done = nullptr;
        bool has_done = false;                                 // ev_done has been recorded: the next call's copies into row[] wait for it
        int peer = 2;                                          // peer access to the source GPU of the last staged batch: 1 granted (direct xGMI copies), 0 refused
                                                               // (the runtime bounces the copies), 2 never needed (same device, or nothing staged yet)
    };
    // One persistent host thread per shard beyond the first (shard 0 runs on the calling thread): created on first use, parked on a
    // condition variable between calls.
    struct ShardWorker {
        std::thread th; std::mutex m; std::condition_variable cv;
        std::function<void()> job; bool busy = false, quit = false;
        void loop() {
            std::unique_lock<std::mutex> lk(m);
            for (;;) {
                cv.wait(lk, [&] { return busy || quit; });
                if (quit) return;
                std::function<void()> j = std::move(job);
                lk.unlock(); j(); lk.lock();
                busy = false; cv.notify_all();
            }
        }
        void submit(std::function<void()> j) { { std::lock_guard<std::mutex> lk(m); job = std::move(j); busy = true; } cv.notify_all(); }
        void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return !busy; }); }
    };
    std::vector<std::unique_ptr<ShardWorker>> workers;
    std::mutex pool_mu;                                        // one batch call at a time hands work to the pool
    std::vector<zkv_ctx*> shards;
    std::vector<ShardDev> sh;
    std::vector<hipEvent_t> ev_in;                       // per source device: "the caller's stream has produced the inputs"
    zkv_ctx* kid[2] = {nullptr, nullptr};
    bool kid_ran[2] = {false, false};        // which sub-batch of the most recent mixed call was non-empty (zkv_ctx_last_stage_ms)
    uint8_t* mx[20] = {nullptr};
    size_t mx_cap[20] = {0};
    std::mutex mu;
};

// Proofs per chunk (= per launch of the stage kernels): ZKV_CHUNK, default 2^20, clamped to [64, 2^26].  The upper clamp is a
// correctness bound, not a tuning choice: the lane-pair kernels address their workspace rows through ONE 32-bit byte offset per lane
// (SoaRef::off = (8 * cap + i) * 4 in k_miller2), which wraps from cap = 2^32 / 36 (about 2^26.8) on; 2^26 proofs per chunk also is
// 248 GB of workspace, i.e. all of one MI355X.
static size_t chunk_capacity() {
    const char* e = getenv("ZKV_CHUNK");
    size_t c = e ? (size_t)strtoull(e, nullptr, 10) : (size_t)1 << 20;      // upper bound: the workspace is sized on demand
    if (c < 64) c = 64;
    if (c > ((size_t)1 << 26)) c = (size_t)1 << 26;
    return (c + 63) & ~(size_t)63;
}
ZKV_EXPORT size_t zkv_chunk_capacity(void) { return chunk_capacity(); }

// Chunks of at most this many proofs take the coefficient-parallel kernels (one proof per 16 lanes).  Measured (RISC Zero,
// tools/small_batch_sweep.sh, profiles/round2_g_small_batch_sweep.txt): 3.4-3.8 ms against 7.8 ms up to 4,096 proofs (one wavefront per
// SIMD), 5.6 against 7.8 ms at 8,192 (two), 7.9 against 7.9 ms from 10,240 on (a second round of wavefronts): above 8,192 the
// lane-pair kernels win because they do a third of the work per proof.
// ZKV_WIDE_BELOW=0 disables the 16-lane kernels.
// Chunks of at most this many proofs run the vk_x stage with one proof per wavefront (k_msm_w: latency instead of throughput).
// ZKV_MSM_WAVE_BELOW=0 disables it.
static size_t msm_wave_below() {
    const char* e = getenv("ZKV_MSM_WAVE_BELOW");
    return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)2048;
}
// Window width of the vk_x stage's fixed-base tables: 16 (default) adds rows of 65,536 entries (4 MB each: SP1 134 MB, RISC Zero 67 MB per
// context) that halve the stage's additions; 8 keeps the L2-resident 8-bit rows only.
static int msm_window_bits() {
    const char* e = getenv("ZKV_MSM_WINDOW_BITS");
    return (e && atoi(e) == 8) ? 8 : 16;
}
static size_t wide_below() {
    const char* e = getenv("ZKV_WIDE_BELOW");
    return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)12288;     // round 4 (profiles/round4_batch_sweep.txt): 16 lanes 6.9-7.2 ms against 7.3 for lane pairs up to here, 8.7 beyond
}
// One lane-pair wavefront holds 32 proofs and a SIMD holds two wavefronts: a launch of up to 32,768 proofs puts one wavefront on
// every SIMD (7.3 ms for the Miller loop and the final exponentiation together), the next 32,768 a second one (10.8 ms for both), and so
// on: T(k layers) grows by 3.5 ms from odd to even k and by 7 ms from even to odd.  A chunk of 32,768 k + r proofs with a small r would
// pay a whole layer for r proofs; instead the last r proofs take the mapping their own number selects (one or two wavefronts per proof,
// 16 lanes per proof: 2-6.4 ms alone):
//  * k odd: the last layer of the others is ONE lane-pair wavefront per SIMD, which issues at 0.74 of the rate of two -- the tail's
//    kernels run BESIDE it on the second stream (234 + 201 VGPRs fit a SIMD together): 33,792 proofs 10.7 -> 7.8 ms (9.7 with the tail
//    after the others), 36,864 10.7 -> 8.0, 40,960 10.7 -> 9.8.  Worth it up to r = ZKV_TAIL_SPLIT_BELOW (default 8,192) for one layer and
//    half of that for three and more, for chunks of up to ZKV_TAIL_SPLIT_MAX proofs (default 2^18: beyond, the odd layer is a few per
//    cent of the launch and the tail disturbs more than it fills);
//  * k even: every SIMD is full, and kernels launched beside would take register space from lane-pair wavefronts at the start and push
//    them into a layer of their own at the end -- the tail runs AFTER the others: 69,632 proofs 18.9 -> 14.2 ms, 135,168 30.0 -> 26.6,
//    263,168 50.3 -> 46.4.  Worth it up to r = ZKV_TAIL_SPLIT_EVEN_BELOW (default 12,288 = the most the 16-lane kernels take: 6.4 ms
//    against the 7 ms of a layer of lone wavefronts), any chunk size.
// ZKV_TAIL_SPLIT_BELOW=0 disables both; ZKV_TAIL_BESIDE=0 runs every tail after the others (profiles/round4_tail_beside.txt).
static bool tail_beside() {
    const char* e = getenv("ZKV_TAIL_BESIDE");
    return !(e && atoi(e) == 0);
}
static size_t tail_split_max() {
    const char* e = getenv("ZKV_TAIL_SPLIT_MAX");
    return e ? (size_t)strtoull(e, nullptr, 10) : ((size_t)1 << 18);
}
static size_t tail_split_below() {
    const char* e = getenv("ZKV_TAIL_SPLIT_BELOW");
    return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)8192;
}
static size_t tail_split_even_below() {
    const char* e = getenv("ZKV_TAIL_SPLIT_EVEN_BELOW");
    return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)12288;
}
// how many proofs at the end of an n-proof chunk take the small-batch mapping (0: none), and whether beside the others or after them
static size_t tail_of_chunk(size_t n, bool* beside) {
    const size_t layer = 32768, k = n / layer, r = n % layer;
    *beside = false;
    if (n <= layer || !r || !tail_split_below()) return 0;
    if (k & 1) { *beside = tail_beside(); return (n <= tail_split_max() && r <= (k == 1 ? tail_split_below() : tail_split_below() / 2)) ? r : 0; }
    return r <= tail_split_even_below() ? r : 0;
}
// The workspace rows of proofs [off, off + ...) of a chunk: word k of proof i sits at base[k * cap + i], so the same capacity with every
// base advanced by `off` elements addresses them as proofs 0, 1, ...
static Workspace ws_from(const Workspace& ws, size_t off) {
    Workspace w = ws;
    w.prep += off; w.norm += off; w.f += off; w.fe += off; w.flags += off; w.g2bad += off;
    return w;
}
// Chunks of at most this many proofs run the Miller loop and the final exponentiation with ONE PROOF PER WAVEFRONT (k_miller_w64 /
// k_finalexp_w64: four slices of 16 lanes; 1,024 proofs are one wavefront on every SIMD of the chip, 2,048 two).  Measured (RISC Zero,
// profiles/round3_e_latency_threshold_sweep.txt), one proof per wavefront against 16 lanes per proof: 1 proof 2.07 / 3.50 ms, 1,024
// proofs 2.56 / 3.65, 1,536 3.03 / 3.69, 2,048 3.61 / 3.78, 3,072 5.07 / 3.71.  ZKV_WAVE_BELOW=0 disables it.
// Chunks of at most this many proofs give the Miller loop TWO wavefronts per proof (k_miller_w64d: one steps the running point and
// tabulates the line coefficients, the other accumulates f).  Measured (RISC Zero, profiles/round3_j_dual_threshold_sweep.txt), two
// wavefronts against one per proof: 1 proof 1.69 / 2.08 ms, 128 proofs 1.90 / 2.15, 256 1.99 / 2.40, 512 2.10 / 2.49, 768 2.47 / 2.56,
// 1,024 (two wavefronts on every SIMD from the Miller kernel alone) 2.76 / 2.55.  ZKV_DUAL_BELOW=0 disables it.
static size_t dual_below() {
    const char* e = getenv("ZKV_DUAL_BELOW");
    return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)768;
}
static size_t wave_below() {
    const char* e = getenv("ZKV_WAVE_BELOW");
    return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)2048;
}

static bool device_is_gfx950(int dev) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return false;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0;
}

ZKV_EXPORT int zkv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (int i = 0; i < n; i++) ok += device_is_gfx950(i) ? 1 : 0;
    return ok;
}
ZKV_EXPORT const char* zkv_version(void) { return "zkv-mi355x 0.1 (gfx950)"; }

#define HIP_TRY(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); return ZKV_ERR_HIP; } } while (0)

// Releases every device resource of the context and returns it to the "not set up" state (safe on a partially set-up context).
static void ctx_free_device(zkv_ctx* c) {
    if (!c->dev_ready && !c->stream) return;
    (void)hipSetDevice(c->device);
    c->m16 = {nullptr, {0, 0, 0, 0, 0}};
    void** ptrs[] = {(void**)&c->d_tab, (void**)&c->d_msm16, (void**)&c->ws.prep, (void**)&c->ws.norm, (void**)&c->ws.f, (void**)&c->ws.fe, (void**)&c->ws.flags,
                     (void**)&c->ws.g2bad, (void**)&c->d_blob, (void**)&c->d_a, (void**)&c->d_b, (void**)&c->d_pv, (void**)&c->d_status,
                     (void**)&c->d_recv, (void**)&c->d_off, (void**)&c->d_pvoff, (void**)&c->d_cd[0], (void**)&c->d_cd[1], (void**)&c->d_kind,
                     (void**)&c->d_cdoff[0], (void**)&c->d_cdoff[1], (void**)&c->d_len, (void**)&c->d_pvlen, (void**)&c->d_st_all,
                     (void**)&c->d_rv_all, (void**)&c->d_inst, (void**)&c->d_inst_idx, (void**)&c->d_pkey, (void**)&c->d_plonk_tab,
                     (void**)&c->d_agg_tab, (void**)&c->d_agg, (void**)&c->ws2.prep, (void**)&c->ws2.norm, (void**)&c->ws2.f, (void**)&c->ws2.fe,
                     (void**)&c->ws2.flags, (void**)&c->ws2.g2bad, (void**)&c->d_status2, (void**)&c->d_agg_cnt, (void**)&c->ws3.prep, (void**)&c->ws3.flags,
                     (void**)&c->ws3.g2bad, (void**)&c->d_status3, (void**)&c->d_agg_idx};
    for (void** p : ptrs) { if (*p) (void)hipFree(*p); *p = nullptr; }
    c->ws2.cap = 0; c->ws3 = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0}; c->agg_cap = 0; c->agg_key_ok = false;
    for (int k = 0; k < 6; k++) { if (c->hb[k]) (void)hipFree(c->hb[k]); c->hb[k] = nullptr; c->hb_cap[k] = 0; }
    for (int k = 0; k < 20; k++) { if (c->mx[k]) (void)hipFree(c->mx[k]); c->mx[k] = nullptr; c->mx_cap[k] = 0; }
    c->ws.cap = 0; c->blob_cap = c->pv_cap = 0; c->cd_cap[0] = c->cd_cap[1] = c->st_all_cap = c->rv_all_cap = 0;
    hipEvent_t* evs[] = {&c->ev[0], &c->ev[1], &c->ev[2], &c->ev[3], &c->ev[4], &c->ev[5], &c->ev_wire[0], &c->ev_wire[1], &c->ev_done,
                         &c->ev_copied[0], &c->ev_copied[1], &c->ev_decoded[0], &c->ev_decoded[1], &c->ev_fork, &c->ev_join, &c->ev_seg[0], &c->ev_seg[1]};
    for (hipEvent_t* e : evs) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
    c->has_done = false; c->wire_timed = false;
    hipStream_t* streams[] = {&c->copy_stream, &c->side, &c->stream};
    for (hipStream_t* st : streams) { if (*st) (void)hipStreamDestroy(*st); *st = nullptr; }
    c->dev_ready = false;
}

// Lazily creates the streams, the VK tables (set-up kernels) and the events; the per-chunk workspace comes from ctx_reserve().
static int ctx_device_setup(zkv_ctx* c) {
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto& e : c->ev) HIP_TRY(hipEventCreate(&e));
    for (auto& e : c->ev_wire) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    HIP_TRY(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    for (auto& e : c->ev_seg) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int b = 0; b < 2; b++) {
        HIP_TRY(hipEventCreateWithFlags(&c->ev_copied[b], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_decoded[b], hipEventDisableTiming));
    }
    if (c->vm != ZKV_VM_BN254 && c->vm != ZKV_VM_MIXED) {
        VkRaw raw;
        if (c->vm == ZKV_VM_RISC0 || c->vm == ZKV_VM_RISC0_SET) host::fill_vk_risc0(raw, c->control_root_0, c->control_root_1, c->control_id);
        else if (c->vm == ZKV_VM_GROTH16) host::fill_vk_generic(raw, c->gvk, c->g_n_ic);
        else if (c->vm == ZKV_VM_SP1_PLONK) {
            // the pairing of a PLONK proof has two FIXED pairs: the SRS's [1]_2 and [tau]_2 take the line-table slots of gamma and
            // delta; there is no (alpha, beta) pair (alpha = infinity contributes 1) and no IC points
            memset(&raw, 0, sizeof raw);
            const int perm[4] = {1, 0, 3, 2};
            for (int k = 0; k < 4; k++) { host::be_to_limbs(raw.gamma[k], c->pk_g2 + 32 * perm[k]); host::be_to_limbs(raw.delta[k], c->pk_g2 + 128 + 32 * perm[k]); }
        }
        else host::fill_vk_sp1(raw);
        if (c->id_ge_r) memset(raw.fixed_scalar[5], 0, 32);      // never used: every proof fails the range check first
        // the raw key and the instance parameters are only read by the set-up kernels; mx[0] / mx[1] hold them until those are done
        HIP_TRY(hipMalloc(&c->mx[0], sizeof(VkRaw)));
        VkRaw* d_raw = (VkRaw*)c->mx[0];
        HIP_TRY(hipMalloc(&c->d_tab, sizeof(VkTables)));
        HIP_TRY(hipMemsetAsync(c->d_tab, 0, sizeof(VkTables), c->stream));
        HIP_TRY(hipMemcpyAsync(d_raw, &raw, sizeof raw, hipMemcpyHostToDevice, c->stream));
        launch_setup(d_raw, c->d_tab, c->stream);
        HIP_TRY(hipGetLastError());
        const bool agg_vm = c->vm == ZKV_VM_RISC0 || c->vm == ZKV_VM_RISC0_SET || c->vm == ZKV_VM_SP1 || c->vm == ZKV_VM_GROTH16;
        if (agg_vm && msm_window_bits() == 16) {
            // 16-bit window rows for the vk_x stage of big batches: 4 MB per row, built from the 8-bit rows k_setup_msm has just written
            Msm16 m = {nullptr, {0, 0, 0, 0, 0}};
            uint32_t rows = 0;
            for (uint32_t b = 0; b < raw.n_var && b < (uint32_t)MAX_VAR; b++) { m.row0[b] = rows; rows += (raw.var_windows[b] + 1) / 2; }
            if (rows && rows <= MSM16_MAX_ROWS) {
                const size_t bytes = (size_t)rows * 65536 * sizeof(G1A);
                if (hipMalloc(&c->d_msm16, bytes) == hipSuccess) {
                    HIP_TRY(hipMemsetAsync(c->d_msm16, 0, bytes, c->stream));
                    launch_setup_msm16(c->d_tab, m, c->d_msm16, rows, c->stream);
                    HIP_TRY(hipGetLastError());
                    m.tab = c->d_msm16;
                    c->m16 = m;
                } else {                                         // no room for the rows (many contexts on one device): the 8-bit walk, same results
                    (void)hipGetLastError();
                    c->d_msm16 = nullptr;
                }
            }
        }
        if (agg_vm || c->vm == ZKV_VM_SP1_PLONK) {
            HIP_TRY(hipMalloc(&c->d_agg_cnt, 3 * sizeof(unsigned long long)));
            HIP_TRY(hipMemsetAsync(c->d_agg_cnt, 0, 3 * sizeof(unsigned long long), c->stream));
        }
        if (agg_vm) {
            HIP_TRY(hipMalloc(&c->d_agg_tab, sizeof(AggTables)));
            launch_setup_agg(d_raw, c->d_tab, c->d_agg_tab, c->stream);
            HIP_TRY(hipGetLastError());
        }
        if (c->vm == ZKV_VM_RISC0_SET) {
            const size_t k = c->inst_raw.size();
            InstConsts ic;
            host::sha256_host((const uint8_t*)"risc0.Groth16ReceiptVerifierParameters", 38, ic.tag);
            host::risc0_vk_digest(ic.vk_digest);
            HIP_TRY(hipMalloc(&c->mx[1], sizeof(InstRaw) * k));
            InstRaw* d_in = (InstRaw*)c->mx[1];
            HIP_TRY(hipMalloc(&c->d_inst, sizeof(InstTab) * k));
            HIP_TRY(hipMemcpyAsync(d_in, c->inst_raw.data(), sizeof(InstRaw) * k, hipMemcpyHostToDevice, c->stream));
            launch_setup_instances(d_raw, ic, d_in, c->d_inst, (uint32_t)k, c->stream);
            HIP_TRY(hipGetLastError());
            c->inst_host.resize(k);
            HIP_TRY(hipMemcpyAsync(c->inst_host.data(), c->d_inst, sizeof(InstTab) * k, hipMemcpyDeviceToHost, c->stream));
        }
        if (c->vm == ZKV_VM_SP1_PLONK) {
            HIP_TRY(hipMalloc(&c->mx[1], sizeof(PlonkKeyRaw)));
            HIP_TRY(hipMalloc(&c->d_pkey, sizeof(PlonkKey)));
            HIP_TRY(hipMemcpyAsync(c->mx[1], &c->pk_raw, sizeof(PlonkKeyRaw), hipMemcpyHostToDevice, c->stream));
            launch_plonk_setup((const PlonkKeyRaw*)c->mx[1], c->d_pkey, c->stream);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int k = 0; k < 2; k++) { if (c->mx[k]) (void)hipFree(c->mx[k]); c->mx[k] = nullptr; }
        uint32_t valid = 0;
        HIP_TRY(hipMemcpy(&valid, &c->d_tab->vk_valid, sizeof valid, hipMemcpyDeviceToHost));
        c->vk_invalid = valid == 0;
        if (c->d_agg_tab) {
            uint32_t ok = 0;
            HIP_TRY(hipMemcpy(&ok, &c->d_agg_tab->ok, sizeof ok, hipMemcpyDeviceToHost));
            c->agg_key_ok = ok != 0;
        }
        if (c->vm == ZKV_VM_SP1_PLONK) c->agg_key_ok = !c->vk_invalid;      // (no (alpha, beta) pair: nothing else to tabulate)
    }
    return ZKV_OK;
}
static int ctx_device_init(zkv_ctx* c) {
    if (c->dev_ready) return hipSetDevice(c->device) == hipSuccess ? ZKV_OK : ZKV_ERR_HIP;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return ZKV_ERR_NO_DEVICE; }
    if (c->device < 0 || c->device >= n || !device_is_gfx950(c->device)) return ZKV_ERR_NO_DEVICE;
    HIP_TRY(hipSetDevice(c->device));
    const int rc = ctx_device_setup(c);
    if (rc != ZKV_OK) { ctx_free_device(c); return rc; }       // a failed set-up leaves nothing behind; the next call starts over
    c->ws.cap = 0;                                       // per-chunk buffers: ctx_reserve()
    c->dev_ready = true;
    return ZKV_OK;
}

// Per-chunk buffers (3.7 KB of workspace per proof in flight) are sized by the largest batch seen so far, rounded up to a power
// of two, at most ZKV_CHUNK (default 2^20) proofs: a context that only ever verifies single proofs stays small, a 2^20-proof
// batch runs as one chunk (larger launches amortise kernel tails: 4.12 M proofs/s at 2^18 per chunk against 4.00 at 2^17).
// Growing frees the old buffers, which synchronises the device, so work in flight on them has finished.
// Buffers of the aggregate check, sized with the workspace (about 0.9 KB per proof in flight on top of its 3.7 KB): 224 B of rows per proof, one
// pseudo-proof workspace per 16 proofs (the smallest sub-batch), and the dense workspace of the second pass (own PREP rows, flags, statuses, index list).
static int agg_reserve(zkv_ctx* c) {
    if (!c->agg_on || !c->agg_key_ok || c->agg_cap >= c->ws.cap) return ZKV_OK;
    void** bufs[] = {(void**)&c->d_agg, (void**)&c->ws2.prep, (void**)&c->ws2.norm, (void**)&c->ws2.f, (void**)&c->ws2.fe, (void**)&c->ws2.flags,
                     (void**)&c->ws2.g2bad, (void**)&c->d_status2, (void**)&c->ws3.prep, (void**)&c->ws3.flags, (void**)&c->ws3.g2bad, (void**)&c->d_status3,
                     (void**)&c->d_agg_idx};
    for (void** b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
    c->agg_cap = 0; c->ws2.cap = 0; c->ws3.cap = 0;
    const size_t cap = c->ws.cap, cap2 = (cap + 15) / 16;      // room for the smallest sub-batch size
    if (hipMalloc(&c->d_agg, sizeof(uint32_t) * WS_AGG_WORDS * cap) != hipSuccess ||
        hipMalloc(&c->ws2.prep, sizeof(uint32_t) * WS_PREP_WORDS * cap2) != hipSuccess ||
        hipMalloc(&c->ws2.norm, sizeof(uint32_t) * WS_NORM_WORDS * cap2) != hipSuccess ||
        hipMalloc(&c->ws2.f, sizeof(uint32_t) * WS_F_WORDS * cap2) != hipSuccess ||
        hipMalloc(&c->ws2.fe, sizeof(uint32_t) * WS_FE_WORDS * cap2) != hipSuccess ||
        hipMalloc(&c->ws2.flags, sizeof(uint32_t) * cap2) != hipSuccess || hipMalloc(&c->ws2.g2bad, sizeof(uint32_t) * cap2) != hipSuccess ||
        hipMalloc(&c->d_status2, cap2) != hipSuccess ||
        hipMalloc(&c->ws3.prep, sizeof(uint32_t) * WS_PREP_WORDS * cap) != hipSuccess || hipMalloc(&c->ws3.flags, sizeof(uint32_t) * cap) != hipSuccess ||
        hipMalloc(&c->ws3.g2bad, sizeof(uint32_t) * cap) != hipSuccess || hipMalloc(&c->d_status3, cap) != hipSuccess ||
        hipMalloc(&c->d_agg_idx, sizeof(uint32_t) * cap) != hipSuccess) {
        // no room for the extra 0.9 KB per proof in flight: the chunk takes the ordinary kernels (enqueue_chunk looks at agg_cap)
        (void)hipGetLastError();
        for (void** b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
        return ZKV_OK;
    }
    c->ws3.norm = c->ws.norm; c->ws3.f = c->ws.f; c->ws3.fe = c->ws.fe; c->ws3.cap = cap;
    c->ws2.cap = cap2; c->agg_cap = cap;
    return ZKV_OK;
}
static int ctx_reserve(zkv_ctx* c, size_t want) {
    const size_t limit = chunk_capacity();
    if (want > limit) want = limit;
    if (want <= c->ws.cap) return agg_reserve(c);
    size_t cap = 4096;
    while (cap < want) cap <<= 1;
    if (cap > limit) cap = limit;
    void** bufs[] = {(void**)&c->ws.prep, (void**)&c->ws.norm, (void**)&c->ws.f, (void**)&c->ws.fe, (void**)&c->ws.flags, (void**)&c->ws.g2bad,
                     (void**)&c->d_a, (void**)&c->d_b, (void**)&c->d_status, (void**)&c->d_recv, (void**)&c->d_off, (void**)&c->d_pvoff,
                     (void**)&c->d_inst_idx, (void**)&c->d_len, (void**)&c->d_pvlen, (void**)&c->d_kind, (void**)&c->d_cdoff[0], (void**)&c->d_cdoff[1],
                     (void**)&c->d_plonk_tab};
    for (void** b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
    c->ws.cap = 0;
    if (hipMalloc(&c->ws.prep, sizeof(uint32_t) * WS_PREP_WORDS * cap) != hipSuccess ||
        hipMalloc(&c->ws.norm, sizeof(uint32_t) * WS_NORM_WORDS * cap) != hipSuccess ||
        hipMalloc(&c->ws.f, sizeof(uint32_t) * WS_F_WORDS * cap) != hipSuccess ||
        hipMalloc(&c->ws.fe, sizeof(uint32_t) * WS_FE_WORDS * cap) != hipSuccess ||
        hipMalloc(&c->ws.flags, sizeof(uint32_t) * cap) != hipSuccess || hipMalloc(&c->ws.g2bad, sizeof(uint32_t) * cap) != hipSuccess ||
        hipMalloc(&c->d_a, 32 * cap) != hipSuccess || hipMalloc(&c->d_b, 32 * cap) != hipSuccess ||
        hipMalloc(&c->d_status, cap) != hipSuccess || hipMalloc(&c->d_recv, 4 * cap) != hipSuccess ||
        hipMalloc(&c->d_off, sizeof(uint64_t) * (cap + 1)) != hipSuccess ||
        hipMalloc(&c->d_pvoff, sizeof(uint64_t) * (cap + 1)) != hipSuccess ||
        hipMalloc(&c->d_inst_idx, sizeof(uint32_t) * cap) != hipSuccess ||
        hipMalloc(&c->d_len, sizeof(uint32_t) * cap) != hipSuccess || hipMalloc(&c->d_pvlen, sizeof(uint32_t) * cap) != hipSuccess ||
        hipMalloc(&c->d_kind, cap) != hipSuccess || hipMalloc(&c->d_cdoff[0], sizeof(uint64_t) * (cap + 1)) != hipSuccess ||
        hipMalloc(&c->d_cdoff[1], sizeof(uint64_t) * (cap + 1)) != hipSuccess ||
        (c->vm == ZKV_VM_SP1_PLONK && hipMalloc(&c->d_plonk_tab, sizeof(uint32_t) * PLONK_TAB_WORDS * cap) != hipSuccess)) {
        (void)hipGetLastError();
        return ZKV_ERR_OOM;
    }
    c->ws.cap = cap;
    return agg_reserve(c);
}
// device set-up + buffers for a batch of n
static int ctx_ready(zkv_ctx* c, size_t n) {
    int rc = ctx_device_init(c);
    return rc != ZKV_OK ? rc : ctx_reserve(c, n ? n : 1);
}
static int grow(uint8_t** p, size_t* cap, size_t need) {
    if (need <= *cap) return ZKV_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    size_t want = need + need / 4 + 4096;
    if (hipMalloc(p, want) != hipSuccess) { (void)hipGetLastError(); return ZKV_ERR_OOM; }
    *cap = want;
    return ZKV_OK;
}

// Cross-stream ordering of consecutive calls on one context (see zkv_ctx::ev_done).
static int order_after_previous(zkv_ctx* c, hipStream_t s) {
    if (c->has_done) HIP_TRY(hipStreamWaitEvent(s, c->ev_done, 0));
    c->agg_look = true;
    return ZKV_OK;
}
static int mark_done(zkv_ctx* c, hipStream_t s) {
    HIP_TRY(hipEventRecord(c->ev_done, s));
    c->has_done = true;
    return ZKV_OK;
}

// Chunks of at least this many proofs take the aggregate check when it is switched on.  Measured (SP1, tools/bench_aggregate.py,
// profiles/round3_t_aggregate_sizes.txt), aggregate / ordinary, all proofs valid: 2^14 7.4 / 8.2 ms, 2^15 8.0 / 8.4, 2^16 10.7 / 13.4,
// 2^17 16.6 / 24.6, 2^18 29.6 / 47.3, 2^20 104 / 183; one proof in 64 rejected (a fifth of those at the pairing): 2^16 19.1 / 13.4,
// 2^17 25.3 / 24.6, 2^18 40.0 / 47.4, 2^20 121 / 182 -- the pseudo-proofs and the second pass over failed sub-batches each cost a
// kernel latency, which small chunks cannot hide.  ZKV_AGG_MIN overrides.
static size_t agg_min() {
    const char* e = getenv("ZKV_AGG_MIN");
    size_t v = e ? (size_t)strtoull(e, nullptr, 10) : (size_t)131072;
    return v < 64 ? 64 : v;
}
// Proofs per Miller accumulator in the aggregate check (k_agg_miller; 1 = k_miller2, an accumulator per proof).  ZKV_AGG_GROUP overrides;
// a sub-batch must hold at least two groups.
static uint32_t agg_group(uint32_t sub) {
    const char* e = getenv("ZKV_AGG_GROUP");
    uint32_t g = e ? (uint32_t)strtoul(e, nullptr, 10) : 4u;
    if (g != 1 && g != 2 && g != 4 && g != 8) g = 4;
    while (g > 1 && sub / g < 2) g >>= 1;
    return g;
}
// Miller loop / final exponentiation of n proofs in workspace ws with the kernel family the chunk size selects (as enqueue_chunk does)
static void launch_miller_by_size(zkv_ctx* c, size_t n, const Workspace& ws, uint8_t* status, hipStream_t s) {
    if (n <= dual_below()) launch_miller_w64d(n, c->d_tab, ws, status, s);
    else if (n <= wave_below()) launch_miller_w64(n, c->d_tab, ws, s);
    else if (n <= wide_below()) launch_miller_w(n, c->d_tab, ws, s);
    else launch_miller2(n, c->d_tab, ws, status, s);
}
static void launch_finalexp_by_size(size_t n, const Workspace& ws, uint8_t* status, hipStream_t s) {
    if (n <= wave_below()) launch_finalexp_w64(n, ws, status, s);
    else if (n <= wide_below()) launch_finalexp_w(n, ws, status, s);
    else launch_finalexp2(n, ws, status, s);
}
// enable = 1 ("automatic"): before a chunk is enqueued, and only if everything enqueued earlier on this context has finished (no
// waiting), the counters tell which fraction of the sub-batches checked since the last look failed; from it the rate p of proofs that
// fail at the pairing, and from p the size for the coming chunks: a failed sub-batch costs its `sub` proofs a second, ordinary
// verification, a sub-batch costs one pseudo-proof: per proof 1 / sub + sub p in units of one verification, least near sub = 1 / sqrt(p).
// Measured (2^20 SP1 proofs): no failures 128 best (87 ms; 91.5 at 64), one proof in 320 failing 16 best (109 ms; 133 at 64).
//
// Switching OFF: at the smallest size a sub-batch that fails costs its 16 proofs the aggregate pass AND the ordinary pass; from about one
// failing sub-batch in three on the check is a net loss (measured: the aggregate pass is ~0.6 of an ordinary one at size 16), and a
// stream of bad proofs would make it a slow-down knob.  Above that rate the following AGG_PAUSE_MIN (then twice as many, up to
// AGG_PAUSE_MAX) chunks run without the check; the chunk after a pause probes again with size 16, and the check stays on once the
// rate has fallen.  Only the automatic mode (enable = 1) does any of this: a caller that fixes the size keeps it.
constexpr uint32_t AGG_PAUSE_MIN = 8, AGG_PAUSE_MAX = 64;
// One look per call, at its start: ev_done is recorded once per call, so a later chunk of the same call could see the event of the
// PREVIOUS call complete while its own earlier chunk is still adding to the two counters.
static void agg_adapt(zkv_ctx* c) {
    if (!c->agg_look) return;
    c->agg_look = false;
    if (!c->agg_auto || !c->has_done || !c->d_agg_cnt || hipEventQuery(c->ev_done) != hipSuccess) { (void)hipGetLastError(); return; }
    unsigned long long v[2];                                   // (on the context's side stream, idle here: waits neither for the caller's streams nor for the
                                                               // host pipeline's segment copies on copy_stream)
    if (hipMemcpyAsync(v, c->d_agg_cnt, sizeof v, hipMemcpyDeviceToHost, c->side) != hipSuccess || hipStreamSynchronize(c->side) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    if (c->agg_resnap) { c->agg_seen[0] = v[0]; c->agg_seen[1] = v[1]; c->agg_resnap = false; return; }
    const unsigned long long checked = v[0] - c->agg_seen[0], failed = v[1] - c->agg_seen[1];
    if (checked < 256) return;                                 // too little to go by
    c->agg_seen[0] = v[0]; c->agg_seen[1] = v[1];
    const double f = (double)failed / (double)checked;         // P(a sub-batch of agg_sub proofs holds a failing proof) = 1 - (1 - p)^sub
    const double p = f >= 1.0 ? 1.0 : 1.0 - pow(1.0 - f, 1.0 / (double)c->agg_sub);
    const uint32_t was = c->agg_sub;
    c->agg_sub = p < 1.0 / 32768 ? 128u : p < 1.0 / 4096 ? 64u : p < 1.0 / 1024 ? 32u : 16u;
    // what size 16 would see with this p: 1 - (1 - p)^16
    const double f16 = was == 16u ? f : 1.0 - pow(1.0 - p, 16.0);
    if (f16 > 1.0 / 3.0) {
        c->agg_pause_len = c->agg_pause_len ? (2 * c->agg_pause_len > AGG_PAUSE_MAX ? AGG_PAUSE_MAX : 2 * c->agg_pause_len) : AGG_PAUSE_MIN;
        c->agg_pause = c->agg_pause_len;
        c->agg_sub = 16u;                                     // the probe after the pause
    } else c->agg_pause_len = 0;
}
// Whether this chunk takes the aggregate check (automatic mode: not during a pause).
static bool agg_wanted(zkv_ctx* c) {
    agg_adapt(c);                                             // (once per call: the decision covers this chunk already)
    if (c->agg_auto && c->agg_pause) { c->agg_pause--; return false; }
    return true;
}
// Fresh coefficients for every chunk (the counter), and a fresh SECRET every AGG_REKEY_CHUNKS chunks when the operating system supplied it
// (zkv_ctx_set_aggregate_check with seed32 = NULL): nothing a long-running service has revealed about old coefficients -- timing, say --
// carries over.  A caller-supplied seed is the caller's responsibility (include/zkv.h).
constexpr uint32_t AGG_REKEY_CHUNKS = 1024;
static uint32_t agg_rekey_chunks() {
    const char* e = getenv("ZKV_AGG_REKEY");                    // tests shorten the interval
    const unsigned long v = e ? strtoul(e, nullptr, 10) : 0;
    return v ? (uint32_t)v : AGG_REKEY_CHUNKS;
}
static void agg_next_coefficients(zkv_ctx* c) {
    if (c->agg_os_seed && ++c->agg_key_age >= agg_rekey_chunks()) {
        uint8_t seed[32];
        if (getrandom(seed, 32, 0) == 32) {
            for (int i = 0; i < 8; i++) c->agg_seed.w[i] = ((uint32_t)seed[4 * i] << 24) | ((uint32_t)seed[4 * i + 1] << 16) | ((uint32_t)seed[4 * i + 2] << 8) | seed[4 * i + 3];
            c->agg_key_age = 0;
        }
        volatile uint8_t* wipe = seed;
        for (int i = 0; i < 32; i++) wipe[i] = 0;
    }
    c->agg_seed.call++;
}
// The aggregate check of one chunk (zkv_agg.h), after PREP: per-proof G1 stage and Miller loop of the variable pair only, one
// pseudo-proof per sub-batch through the ordinary Miller loop and final exponentiation, then the ordinary stages once more for the
// proofs of sub-batches that failed, gathered into a dense workspace (the launches cover the whole chunk -- the host does not know how
// many there are -- and wavefronts past the end of the list leave at once).
static void enqueue_agg(zkv_ctx* c, const PrepArgs& a, hipStream_t s, bool timed) {
    agg_adapt(c);
    const uint32_t sub = c->agg_sub, sub64 = sub < 64 ? sub : 64;        // sub-batches of 128 / 256 proofs are summed per 64-proof block first
    const size_t n64 = (a.n + 63) / 64;
    const size_t n2 = sub > 64 ? (a.n + sub - 1) / sub : n64 * (64 / sub);      // the last block counted in full (empty sub-batches switch themselves off)
    // proofs per Miller accumulator: ZKV_AGG_GROUP = 1 (k_miller2), 2, 4 or 8 (k_agg_miller); at most the sub-batch's eighth... see agg_group()
    const uint32_t grp = agg_group(sub64);
    const InstTab* inst = a.inst ? c->d_inst : nullptr;
    agg_next_coefficients(c);
    // vk_x through summed scalars: one key (no per-proof base) and at most two per-proof signals
    const bool sums = !inst && (c->vm == ZKV_VM_RISC0 || c->vm == ZKV_VM_SP1 || (c->vm == ZKV_VM_GROTH16 && c->g_n_ic >= 1 && c->g_n_ic - 1 <= (uint32_t)AGG_SUM_VARS));
    launch_agg_g1(a.n, c->d_tab, inst, c->ws, c->d_agg, c->agg_seed, sums, s);
    if (timed) { (void)hipEventRecord(c->ev[2], s); (void)hipEventRecord(c->ev[3], s); }
    if (grp > 1) launch_agg_miller(a.n, grp, c->d_tab, c->ws, a.status, s);
    else launch_miller2(a.n, c->d_tab, c->ws, a.status, s);
    launch_agg_reduce(a.n, sub64, sums, grp, c->d_tab, c->ws, c->d_agg, c->d_agg_tab, c->ws2, c->d_status2, sub > 64, s);
    if (sub > 64) launch_agg_combine(n64, n2, sub / 64, c->d_agg_tab, c->ws2, c->d_status2, s);
    launch_miller_by_size(c, n2, c->ws2, c->d_status2, s);
    if (timed) (void)hipEventRecord(c->ev[4], s);
    launch_agg_fprod(a.n, n2, sub, grp, c->ws, c->d_agg, c->ws2, s);
    launch_finalexp_by_size(n2, c->ws2, c->d_status2, s);
    launch_agg_mark(a.n, sub, grp, c->ws, c->d_agg, c->d_status2, a.status, c->d_agg_cnt, c->d_agg_idx, s);
    launch_agg_gather(a.n, c->ws, c->d_agg, c->d_agg_cnt, c->d_agg_idx, c->ws3, c->d_status3, s);
    launch_msm(a.n, c->d_tab, c->m16, inst, c->ws3, s);
    launch_miller2(a.n, c->d_tab, c->ws3, c->d_status3, s);
    launch_finalexp2(a.n, c->ws3, c->d_status3, s);
    launch_agg_scatter(a.n, c->d_agg_cnt, c->d_agg_idx, c->d_status3, a.status, s);
    if (timed) (void)hipEventRecord(c->ev[5], s);
}

// The aggregate check of a PLONK chunk, after the unchanged PREP stage (k_agg_plonk_g1 explains why there is no per-proof Miller loop).
static void enqueue_agg_plonk(zkv_ctx* c, const PrepArgs& a, hipStream_t s, bool timed) {
    agg_adapt(c);
    const uint32_t sub = c->agg_sub, sub64 = sub < 64 ? sub : 64;
    const size_t n64 = (a.n + 63) / 64;
    const size_t n2 = sub > 64 ? (a.n + sub - 1) / sub : n64 * (64 / sub);
    agg_next_coefficients(c);
    launch_agg_plonk_g1(a.n, c->ws, c->d_agg, c->agg_seed, s);
    if (timed) { (void)hipEventRecord(c->ev[2], s); (void)hipEventRecord(c->ev[3], s); }
    launch_agg_reduce(a.n, sub64, false, 1, c->d_tab, c->ws, c->d_agg, nullptr, c->ws2, c->d_status2, sub > 64, s);
    if (sub > 64) launch_agg_combine(n64, n2, sub / 64, nullptr, c->ws2, c->d_status2, s);
    launch_miller_by_size(c, n2, c->ws2, c->d_status2, s);
    if (timed) (void)hipEventRecord(c->ev[4], s);
    launch_finalexp_by_size(n2, c->ws2, c->d_status2, s);
    launch_agg_mark(a.n, sub, 1, c->ws, c->d_agg, c->d_status2, a.status, c->d_agg_cnt, c->d_agg_idx, s);
    launch_agg_plonk_norm(a.n, c->ws, c->d_agg, c->d_agg_cnt, c->d_agg_idx, c->ws3, c->d_status3, s);
    launch_miller2(a.n, c->d_tab, c->ws3, c->d_status3, s);
    launch_finalexp2(a.n, c->ws3, c->d_status3, s);
    launch_agg_scatter(a.n, c->d_agg_cnt, c->d_agg_idx, c->d_status3, a.status, s);
    if (timed) (void)hipEventRecord(c->ev[5], s);
}

// Enqueues the five stages for one chunk (all pointers device-resident).
static void enqueue_chunk(zkv_ctx* c, const PrepArgs& a, hipStream_t s, bool timed) {
    if (timed) (void)hipEventRecord(c->ev[0], s);
    if (c->vm == ZKV_VM_SP1_PLONK) {
        // PLONK: the prep stage does everything up to the two G1 points of the final check (transcript, scalar algebra, MSMs);
        // no per-proof G2 point, so no subgroup check, and no vk_x stage
        PrepArgs ap = a;
        ap.plonk_tab = c->d_plonk_tab;
        launch_plonk_prep(ap, c->d_pkey, c->ws, s);
        if (c->agg_on && c->agg_key_ok && c->agg_cap >= c->ws.cap && c->lanes == 0 && a.n >= agg_min() && agg_wanted(c)) {
            if (timed) (void)hipEventRecord(c->ev[1], s);
            enqueue_agg_plonk(c, a, s, timed);
            return;
        }
        if (timed) { (void)hipEventRecord(c->ev[1], s); (void)hipEventRecord(c->ev[2], s); (void)hipEventRecord(c->ev[3], s); }
        const int pl = c->lanes ? c->lanes : 2;
        const bool wave_p = pl == 64 || pl == 128 || (c->lanes == 0 && a.n <= wave_below());      // (no variable pair: nothing for a second wavefront to do)
        const bool wide_p = wave_p || pl == 16 || (c->lanes == 0 && a.n <= wide_below());
        if (wave_p) launch_miller_w64(a.n, c->d_tab, c->ws, s); else if (wide_p) launch_miller_w(a.n, c->d_tab, c->ws, s); else launch_miller2(a.n, c->d_tab, c->ws, a.status, s);
        if (timed) (void)hipEventRecord(c->ev[4], s);
        if (wave_p) launch_finalexp_w64(a.n, c->ws, a.status, s); else if (wide_p) launch_finalexp_w(a.n, c->ws, a.status, s); else launch_finalexp2(a.n, c->ws, a.status, s);
        if (timed) (void)hipEventRecord(c->ev[5], s);
        return;
    }
    if (c->vm == ZKV_VM_RISC0 || c->vm == ZKV_VM_RISC0_SET) launch_prep_risc0(a, c->consts, c->ws, s);
    else if (c->vm == ZKV_VM_GROTH16) launch_prep_groth16(a, c->ws, s);
    else launch_prep_sp1(a, c->ws, s);
    if (timed) (void)hipEventRecord(c->ev[1], s);
    if (c->agg_on && c->agg_key_ok && c->agg_cap >= c->ws.cap && c->lanes == 0 && a.n >= agg_min() && agg_wanted(c)) { enqueue_agg(c, a, s, timed); return; }
    bool tail_runs_beside = false;
    const size_t tail = c->lanes == 0 ? tail_of_chunk(a.n, &tail_runs_beside) : 0;
    if (tail) {
        // all proofs through the vk_x stage, then the Miller loops and final exponentiations of the first a.n - tail proofs on lane pairs and
        // of the last `tail` through their own small-batch mapping (whose Miller kernels leave the subgroup test of B to k_g2chk2)
        const size_t head = a.n - tail;
        const Workspace wt = ws_from(c->ws, head);
        launch_msm(a.n, c->d_tab, c->m16, a.inst ? c->d_inst : nullptr, c->ws, s);
        if (timed) (void)hipEventRecord(c->ev[2], s);
        // the tail's kernels on the second stream beside the lane-pair kernels of the others (odd number of layers), or after them
        // (tail_of_chunk); disjoint workspace rows and status bytes either way
        const bool beside = tail_runs_beside && c->side != nullptr;
        hipStream_t st = beside ? c->side : s;
        if (beside) { (void)hipEventRecord(c->ev_fork, s); (void)hipStreamWaitEvent(c->side, c->ev_fork, 0); }
        launch_g2chk2(tail, wt, a.status + head, st);
        if (timed) (void)hipEventRecord(c->ev[3], s);
        launch_miller2(head, c->d_tab, c->ws, a.status, s);
        launch_miller_by_size(c, tail, wt, a.status + head, st);
        if (timed) (void)hipEventRecord(c->ev[4], s);
        launch_finalexp2(head, c->ws, a.status, s);
        launch_finalexp_by_size(tail, wt, a.status + head, st);
        if (beside) { (void)hipEventRecord(c->ev_join, c->side); (void)hipStreamWaitEvent(s, c->ev_join, 0); }
        if (timed) (void)hipEventRecord(c->ev[5], s);
        return;
    }
    const int lanes = c->lanes ? c->lanes : 2;       // 2 = one proof per lane pair; 16 = one proof per 16 lanes (small chunks); 64 = per wavefront (smallest)
    const bool dual = lanes == 128 || (c->lanes == 0 && a.n <= dual_below());       // 128 = two wavefronts per proof in the Miller loop
    const bool wave = dual || lanes == 64 || (c->lanes == 0 && a.n <= wave_below());
    const bool wide = wave || lanes == 16 || (c->lanes == 0 && a.n <= wide_below());
    // Lane-pair kernels: the Miller loop itself is the subgroup test of B (miller_loop_p), there is no separate check; stage time
    // [2] is then 0.  16-lane kernels (small chunks, most of the chip idle): the check (k_g2chk2) only needs the PREP output and only
    // its verdict (ws.g2bad; the MSM owns ws.flags) is needed, by the final exponentiation, so it runs on a second stream beside the
    // MSM and the Miller loop, which is computed speculatively for the rare proof whose B fails the check (-0.7 ms).
    const bool fork = wide && c->side != nullptr;
    if (fork) {
        (void)hipEventRecord(c->ev_fork, s);
        (void)hipStreamWaitEvent(c->side, c->ev_fork, 0);
        launch_g2chk2(a.n, c->ws, a.status, c->side);
        (void)hipEventRecord(c->ev_join, c->side);
    }
    if (a.n <= msm_wave_below()) launch_msm_w(a.n, c->d_tab, a.inst ? c->d_inst : nullptr, c->ws, s);     // one proof per wavefront
    else launch_msm(a.n, c->d_tab, c->m16, a.inst ? c->d_inst : nullptr, c->ws, s);
    if (timed) (void)hipEventRecord(c->ev[2], s);
    if (wide && !fork) launch_g2chk2(a.n, c->ws, a.status, s);
    if (timed) (void)hipEventRecord(c->ev[3], s);
    if (dual) launch_miller_w64d(a.n, c->d_tab, c->ws, a.status, s);
    else if (wave) launch_miller_w64(a.n, c->d_tab, c->ws, s);
    else if (wide) launch_miller_w(a.n, c->d_tab, c->ws, s);
    else launch_miller2(a.n, c->d_tab, c->ws, a.status, s);
    if (timed) (void)hipEventRecord(c->ev[4], s);
    if (fork) (void)hipStreamWaitEvent(s, c->ev_join, 0);     // the final exponentiation reads the verdict of the subgroup check
    if (wave) launch_finalexp_w64(a.n, c->ws, a.status, s);
    else if (wide) launch_finalexp_w(a.n, c->ws, a.status, s);
    else launch_finalexp2(a.n, c->ws, a.status, s);
    if (timed) (void)hipEventRecord(c->ev[5], s);
}

// Offsets handed over by the caller must be non-decreasing: the kernels index the blob with them.
static bool offsets_ok(const uint64_t* off, size_t n) {
    for (size_t i = 0; i < n; i++) if (off[i + 1] < off[i]) return false;
    return true;
}
static uint32_t be32_of(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// Host-pointer batch driver shared by risc0 verify / verify_integrity / sp1 verify_proof.
//
// The whole batch (at most 2^22 proofs per pass) is staged in HBM; the copies run on the context's copy stream SEGMENT BY SEGMENT
// while the previous segment is being verified: a short first segment (2^16 proofs: its 25 MB cross PCIe in about a millisecond),
// then segments of up to one workspace (2^20 proofs) whose H2D time hides behind the kernels of the segment before.  Statuses
// stay on the device until the pass is done.  Only the first segment's copy and the status D2H are exposed.
static size_t host_first_segment(size_t n, size_t cap) {
    const char* e = getenv("ZKV_HOST_FIRST_SEGMENT");
    size_t v = e ? (size_t)strtoull(e, nullptr, 10) : (size_t)1 << 16;
    if (v < 64) v = 64;
    if (n <= 2 * v) v = n;
    return v < cap ? v : cap;
}
static int run_host_batch(zkv_ctx* c, size_t n, const uint8_t* blob, const uint64_t* off, const uint8_t* in_a, const uint8_t* in_b,
                          const uint8_t* pv_blob, const uint64_t* pv_off, uint8_t* status, uint8_t* recv) {
    if (!c || (n && (!blob || !off || !status || !in_a))) return ZKV_ERR_INVALID_ARG;
    if (n && (!offsets_ok(off, n) || (pv_off && !offsets_ok(pv_off, n)))) return ZKV_ERR_INVALID_ARG;
    if (recv) memset(recv, 0, 4 * n);
    if (c->vm == ZKV_VM_RISC0 && !c->initialized) {              // risc0/verifier.rs:84-86, 99-101
        memset(status, ZKV_STATUS_INVALID_INITIALIZATION, n);
        return ZKV_OK;
    }
    if (!n) return ZKV_OK;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_ready(c, n);
    if (rc != ZKV_OK) return rc;
    const char* pe = getenv("ZKV_HOST_PASS");                   // proofs staged per pass (tests shrink it to reach the multi-pass loop)
    const size_t cap = c->ws.cap, pass_max = pe && strtoull(pe, nullptr, 10) ? (size_t)strtoull(pe, nullptr, 10) : (size_t)1 << 22;
    const bool sp1 = c->vm == ZKV_VM_SP1 || c->vm == ZKV_VM_SP1_PLONK;
    if ((rc = order_after_previous(c, c->stream)) != ZKV_OK) return rc;
    std::vector<uint64_t> rel, prel;
    for (size_t p0 = 0; p0 < n; p0 += pass_max) {
        const size_t pn = n - p0 < pass_max ? n - p0 : pass_max;
        const uint64_t s0 = off[p0], sbytes = off[p0 + pn] - s0;
        const uint64_t v0 = sp1 ? pv_off[p0] : 0, vbytes = sp1 ? pv_off[p0 + pn] - v0 : 0;
        // growing a buffer frees the old one, which synchronises the device: everything is sized before the first copy
        if ((rc = grow(&c->hb[0], &c->hb_cap[0], (size_t)sbytes + 8)) != ZKV_OK || (rc = grow(&c->hb[1], &c->hb_cap[1], 8 * (pn + 1))) != ZKV_OK ||
            (rc = grow(&c->hb[2], &c->hb_cap[2], 32 * pn)) != ZKV_OK || (in_b && (rc = grow(&c->hb[3], &c->hb_cap[3], 32 * pn)) != ZKV_OK) ||
            (sp1 && ((rc = grow(&c->hb[4], &c->hb_cap[4], (size_t)vbytes + 8)) != ZKV_OK || (rc = grow(&c->hb[5], &c->hb_cap[5], 8 * (pn + 1))) != ZKV_OK)) ||
            (rc = grow(&c->d_st_all, &c->st_all_cap, pn)) != ZKV_OK || (rc = grow(&c->d_rv_all, &c->rv_all_cap, 4 * pn)) != ZKV_OK) return rc;
        // offsets relative to the pass (the caller's array is used as it is when the pass starts at offset 0)
        const uint64_t* o = off + p0; const uint64_t* po = sp1 ? pv_off + p0 : nullptr;
        if (s0) { rel.resize(pn + 1); for (size_t i = 0; i <= pn; i++) rel[i] = off[p0 + i] - s0; o = rel.data(); }
        if (sp1 && v0) { prel.resize(pn + 1); for (size_t i = 0; i <= pn; i++) prel[i] = pv_off[p0 + i] - v0; po = prel.data(); }
        const size_t first = host_first_segment(pn, cap);
        // ZKV_HOST_TRACE=1 (diagnostic): events around the segments' copies and kernels, printed to stderr after the pass
        const bool trace = getenv("ZKV_HOST_TRACE") != nullptr;
        std::vector<hipEvent_t> tev;
        auto mark = [&](hipStream_t st) { if (!trace) return; hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, st); tev.push_back(e); } };
        mark(c->stream);
        for (size_t base = 0, k = 0; base < pn; k++) {
            // (a third, intermediate segment of 2 * first was measured and dropped: 202.6 against 200.7 ms per 2^20 SP1 proofs,
            // alternating runs on one box)
            const size_t m = k == 0 ? first : (pn - base < cap ? pn - base : cap);
            hipStream_t cs = c->copy_stream;
            HIP_TRY(hipMemcpyAsync(c->hb[1] + 8 * base, o + base, 8 * (m + 1), hipMemcpyHostToDevice, cs));
            if (o[base + m] > o[base]) HIP_TRY(hipMemcpyAsync(c->hb[0] + o[base], blob + s0 + o[base], (size_t)(o[base + m] - o[base]), hipMemcpyHostToDevice, cs));
            HIP_TRY(hipMemcpyAsync(c->hb[2] + 32 * base, in_a + 32 * (p0 + base), 32 * m, hipMemcpyHostToDevice, cs));
            if (in_b) HIP_TRY(hipMemcpyAsync(c->hb[3] + 32 * base, in_b + 32 * (p0 + base), 32 * m, hipMemcpyHostToDevice, cs));
            if (sp1) {
                HIP_TRY(hipMemcpyAsync(c->hb[5] + 8 * base, po + base, 8 * (m + 1), hipMemcpyHostToDevice, cs));
                if (po[base + m] > po[base]) HIP_TRY(hipMemcpyAsync(c->hb[4] + po[base], pv_blob + v0 + po[base], (size_t)(po[base + m] - po[base]), hipMemcpyHostToDevice, cs));
            }
            HIP_TRY(hipEventRecord(c->ev_seg[k & 1], cs));
            mark(cs);
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_seg[k & 1], 0));
            PrepArgs a;
            memset(&a, 0, sizeof a);
            a.n = m; a.blob = c->hb[0]; a.off = (const uint64_t*)c->hb[1] + base; a.stride = 0;
            a.in32_a = c->hb[2] + 32 * base; a.in32_b = in_b ? c->hb[3] + 32 * base : nullptr;
            if (sp1) {
                a.pv_blob = c->hb[4]; a.pv_off = (const uint64_t*)c->hb[5] + base;
                a.selector_be = be32_of(c->vm == ZKV_VM_SP1_PLONK ? c->plonk_hash : host::SP1_VERIFIER_HASH);
                a.force_fail = c->vm == ZKV_VM_SP1_PLONK && c->vk_invalid ? 1u : 0u;
            } else {
                a.selector_be = be32_of(c->selector);
                a.force_fail = c->id_ge_r ? 1u : 0u;
            }
            a.status = c->d_st_all + base; a.recv = c->d_rv_all + 4 * base;
            base += m;
            enqueue_chunk(c, a, c->stream, base >= pn);
            mark(c->stream);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemcpyAsync(status + p0, c->d_st_all, pn, hipMemcpyDeviceToHost, c->stream));
        if (recv) HIP_TRY(hipMemcpyAsync(recv + 4 * p0, c->d_rv_all, 4 * pn, hipMemcpyDeviceToHost, c->stream));
        mark(c->stream);
        HIP_TRY(hipStreamSynchronize(c->stream));            // also: the staging buffers are free for the next pass
        if (trace && tev.size() > 1) {
            fprintf(stderr, "zkv host pass of %zu proofs (first segment %zu): events [start, (copy done, kernels done) per segment, statuses back], ms since start:", pn, first);
            for (size_t i = 1; i < tev.size(); i++) { float ms = 0; (void)hipEventElapsedTime(&ms, tev[0], tev[i]); fprintf(stderr, " %.2f", ms); }
            fprintf(stderr, "\n");
        }
        for (auto e : tev) (void)hipEventDestroy(e);
    }
    return ZKV_OK;
}

// Device-pointer fast path (fixed stride), asynchronous on `stream`.
static int run_dev_batch(zkv_ctx* c, size_t n, const uint8_t* d_blob, const uint8_t* d_a, const uint8_t* d_b, const uint8_t* d_pv,
                         size_t pv_len, uint8_t* d_status, uint8_t* d_recv, void* stream) {
    if (!c || (n && (!d_blob || !d_a || !d_status))) return ZKV_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_ready(c, n);
    if (rc != ZKV_OK) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (c->vm == ZKV_VM_RISC0 && !c->initialized) {
        HIP_TRY(hipMemsetAsync(d_status, ZKV_STATUS_INVALID_INITIALIZATION, n, s));
        if (d_recv) HIP_TRY(hipMemsetAsync(d_recv, 0, 4 * n, s));
        return ZKV_OK;
    }
    size_t cap = c->ws.cap;
    if (n && (rc = order_after_previous(c, s)) != ZKV_OK) return rc;
    for (size_t base = 0; base < n; base += cap) {
        size_t m = n - base < cap ? n - base : cap;
        PrepArgs a;
        memset(&a, 0, sizeof a);
        const size_t rec = c->vm == ZKV_VM_SP1_PLONK ? (size_t)ZKV_PLONK_PROOF_BYTES : (size_t)ZKV_SEAL_BYTES;
        a.n = m; a.blob = d_blob + base * rec; a.off = nullptr; a.stride = (uint32_t)rec;
        a.in32_a = d_a + 32 * base; a.in32_b = d_b ? d_b + 32 * base : nullptr;
        if (c->vm == ZKV_VM_SP1 || c->vm == ZKV_VM_SP1_PLONK) {
            a.pv_blob = d_pv + base * pv_len; a.pv_off = nullptr; a.pv_stride = (uint32_t)pv_len;
            a.selector_be = be32_of(c->vm == ZKV_VM_SP1_PLONK ? c->plonk_hash : host::SP1_VERIFIER_HASH);
            a.force_fail = c->vm == ZKV_VM_SP1_PLONK && c->vk_invalid ? 1u : 0u;
        } else {
            a.selector_be = be32_of(c->selector);
            a.force_fail = c->id_ge_r ? 1u : 0u;
        }
        a.status = d_status + base; a.recv = d_recv ? d_recv + 4 * base : nullptr;
        enqueue_chunk(c, a, s, base + cap >= n);
    }
    HIP_TRY(hipGetLastError());
    return n ? mark_done(c, s) : ZKV_OK;
}

// Stage pipeline over m compact records that a device-side front end produced (fixed 260-byte stride plus the true length,
// 32-byte inputs, public values as (start, length) into one blob): the output format of the calldata decoder and of the
// mixed-batch demultiplexer.  Asynchronous on `s`.
static int run_records(zkv_ctx* c, size_t m_total, const uint8_t* seals, const uint32_t* len, const uint8_t* in_a, const uint8_t* in_b,
                       const uint8_t* pv_blob, const uint64_t* pv_start, const uint32_t* pv_len, uint8_t* st, uint8_t* rv, hipStream_t s) {
    if (!m_total) return ZKV_OK;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_ready(c, m_total);
    if (rc != ZKV_OK) return rc;
    if ((rc = order_after_previous(c, s)) != ZKV_OK) return rc;
    const size_t cap = c->ws.cap;
    for (size_t base = 0; base < m_total; base += cap) {
        const size_t m = m_total - base < cap ? m_total - base : cap;
        PrepArgs a;
        memset(&a, 0, sizeof a);
        a.n = m; a.blob = seals + base * ZKV_SEAL_BYTES; a.stride = ZKV_SEAL_BYTES; a.len = len + base;
        a.in32_a = in_a + 32 * base;
        if (c->vm == ZKV_VM_SP1) {
            a.pv_blob = pv_blob; a.pv_off = pv_start + base; a.pv_len = pv_len + base;
            a.selector_be = be32_of(host::SP1_VERIFIER_HASH);
        } else {
            a.in32_b = in_b + 32 * base;
            a.selector_be = be32_of(c->selector);
            a.force_fail = c->id_ge_r ? 1u : 0u;
        }
        a.status = st + base; a.recv = rv ? rv + 4 * base : nullptr;
        enqueue_chunk(c, a, s, base + cap >= m_total);
    }
    HIP_TRY(hipGetLastError());
    return mark_done(c, s);
}

// ------------------------------------------------------------------ sharded (multi-device) contexts (SURVEY 8b `device_mask`, 8e)
// One single-device context per shard behind the ordinary batch entry points; proofs are independent, so a batch splits into
// contiguous ranges and nothing is exchanged between shards.  Host-buffer batches: one host thread per shard calls the single-device
// entry point on its range of the caller's buffers, so every GPU pulls its rows over its own PCIe link (no bounce through a root
// GPU).  Device-resident batches: rows are copied from the GPU that holds them to each shard's GPU with hipMemcpyPeerAsync (one direct
// xGMI link per peer) in two pieces -- the second piece travels behind the first piece's kernels -- and the statuses return the same way.
static inline bool is_sharded(const zkv_ctx* c) { return c && !c->shards.empty(); }
static size_t env_size(const char* name, size_t dflt) {
    const char* e = getenv(name);
    return e && *e ? (size_t)strtoull(e, nullptr, 10) : dflt;
}
// shards that get work: at least ZKV_SHARD_MIN (default 1,024) proofs each -- a single proof stays on one GPU
static size_t shards_used(const zkv_ctx* c, size_t n) {
    size_t mn = env_size("ZKV_SHARD_MIN", 1024);
    if (mn < 1) mn = 1;
    size_t k = (n + mn - 1) / mn;
    if (k < 1) k = 1;
    return k < c->shards.size() ? k : c->shards.size();
}
static inline void shard_range(size_t n, size_t used, size_t k, size_t* lo, size_t* hi) {      // contiguous, remainder to the low shards
    const size_t q = n / used, r = n % used;
    *lo = k * q + (k < r ? k : r);
    *hi = *lo + q + (k < r ? 1 : 0);
}
// per_shard(child, lo, hi) -> ZKV_*; shard 0 runs on the calling thread
// fn(k) for the shards 1 .. used - 1 on their persistent workers and for shard 0 on the calling thread; returns when all are done.
template <class F> static void shard_parallel(zkv_ctx* c, size_t used, F fn) {
    std::lock_guard<std::mutex> pool(c->pool_mu);
    while (c->workers.size() + 1 < used) {
        std::unique_ptr<zkv_ctx::ShardWorker> w(new zkv_ctx::ShardWorker());
        try { w->th = std::thread([p = w.get()] { p->loop(); }); } catch (const std::system_error&) { break; }     // no thread to be had: that shard runs here
        c->workers.push_back(std::move(w));
    }
    const size_t pooled = c->workers.size() + 1 < used ? c->workers.size() + 1 : used;
    for (size_t k = 1; k < pooled; k++) c->workers[k - 1]->submit([&fn, k] { fn(k); });
    fn(0);
    for (size_t k = pooled; k < used; k++) fn(k);
    for (size_t k = 1; k < pooled; k++) c->workers[k - 1]->wait();
}
static void shard_workers_stop(zkv_ctx* c) {
    for (auto& w : c->workers) {
        { std::lock_guard<std::mutex> lk(w->m); w->quit = true; }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
    }
    c->workers.clear();
}
template <class F> static int run_sharded(zkv_ctx* c, size_t n, F per_shard) {
    const size_t used = shards_used(c, n);
    std::vector<int> rc(used, ZKV_OK);
    shard_parallel(c, used, [&](size_t k) { size_t lo, hi; shard_range(n, used, k, &lo, &hi); rc[k] = per_shard(c->shards[k], lo, hi); });
    for (int r : rc) if (r != ZKV_OK) return r;
    return ZKV_OK;
}
static int shard_dev_setup(zkv_ctx::ShardDev& d, int device) {
    if (d.run) return ZKV_OK;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&d.run, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&d.copy, hipStreamNonBlocking));
    for (auto& e : d.ev_piece) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&d.ev_done, hipEventDisableTiming));
    return ZKV_OK;
}
struct DevRow { const uint8_t* p; size_t stride; };
// child_call(child, m, rows[4] (device pointers on the child's GPU), status, recv (may be null), stream) -> ZKV_*
template <class F>
static int run_sharded_dev(zkv_ctx* c, size_t n, const DevRow* rows, int n_rows, uint8_t* d_status, uint8_t* d_recv, void* stream, F child_call) {
    if (!n) return ZKV_OK;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, rows[0].p) != hipSuccess) { (void)hipGetLastError(); return ZKV_ERR_INVALID_ARG; }
    const int sdev = at.device;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (sdev < 0 || sdev >= ndev) return ZKV_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->ev_in.size() < (size_t)ndev) c->ev_in.resize(ndev, nullptr);
    hipStream_t caller = (hipStream_t)stream;
    if (caller) {                                     // the shards' streams start after what the caller's stream has enqueued so far
        HIP_TRY(hipSetDevice(sdev));
        if (!c->ev_in[sdev]) HIP_TRY(hipEventCreateWithFlags(&c->ev_in[sdev], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->ev_in[sdev], caller));
    }
    const size_t used = shards_used(c, n);
    const bool force = env_size("ZKV_SHARD_FORCE_STAGING", 0) != 0;         // tests: take the peer-copy path on a one-GPU box
    // first piece of a staged range: ZKV_SHARD_FIRST_PIECE when set, else half the range for ranges of at least 2^17 proofs (small
    // launches run the stage kernels below their rate, which costs more than the transfer a smaller first piece would hide)
    const size_t first_env = env_size("ZKV_SHARD_FIRST_PIECE", 0);
    std::vector<int> rc(used, ZKV_OK);
    auto one = [&](size_t k) -> int {
        zkv_ctx* kid = c->shards[k];
        zkv_ctx::ShardDev& d = c->sh[k];
        size_t lo, hi; shard_range(n, used, k, &lo, &hi);
        const size_t m = hi - lo;
        if (!m) return ZKV_OK;
        int r = shard_dev_setup(d, kid->device);
        if (r != ZKV_OK) return r;
        HIP_TRY(hipSetDevice(kid->device));
        const bool staged = force || kid->device != sdev;
        if (caller) { HIP_TRY(hipStreamWaitEvent(d.run, c->ev_in[sdev], 0)); HIP_TRY(hipStreamWaitEvent(d.copy, c->ev_in[sdev], 0)); }
        if (staged) {
            if (kid->device != sdev) {                             // direct xGMI copies where the platform allows; the verdict is kept for zkv_ctx_shard_peer_access
                const hipError_t pe = hipDeviceEnablePeerAccess(sdev, 0);
                (void)hipGetLastError();
                d.peer = (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) ? 1 : 0;
            }
            // the staging rows are about to be overwritten: not before the previous call's kernels on this shard have read them (that call may
            // have been enqueued on another caller stream, or with none, and still be running)
            if (d.has_done) HIP_TRY(hipStreamWaitEvent(d.copy, d.ev_done, 0));
            for (int j = 0; j < n_rows; j++) if ((r = grow(&d.row[j], &d.row_cap[j], m * rows[j].stride + 8)) != ZKV_OK) return r;
            if ((r = grow(&d.st, &d.st_cap, m)) != ZKV_OK || (d_recv && (r = grow(&d.rv, &d.rv_cap, 4 * m)) != ZKV_OK)) return r;
        }
        // two pieces: the first (at most `first` proofs) is what the kernels wait for, the rest travels behind its kernels
        size_t pc[3] = {0, m, m};
        int np = 1;
        const size_t first = first_env ? first_env : (m >= ((size_t)1 << 17) ? m / 2 : 0);
        if (staged && first && m >= 2 * first) { pc[1] = first; np = 2; }
        if (staged) {
            for (int q = 0; q < np; q++) {
                for (int j = 0; j < n_rows; j++)
                    HIP_TRY(hipMemcpyPeerAsync(d.row[j] + pc[q] * rows[j].stride, kid->device, rows[j].p + (lo + pc[q]) * rows[j].stride, sdev,
                                               (pc[q + 1] - pc[q]) * rows[j].stride, d.copy));
                HIP_TRY(hipEventRecord(d.ev_piece[q], d.copy));
            }
        }
        for (int q = 0; q < np; q++) {
            const uint8_t* rp[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int j = 0; j < n_rows; j++) rp[j] = (staged ? d.row[j] : rows[j].p + lo * rows[j].stride) + pc[q] * rows[j].stride;
            if (staged) HIP_TRY(hipStreamWaitEvent(d.run, d.ev_piece[q], 0));
            uint8_t* st = (staged ? d.st : d_status + lo) + pc[q];
            uint8_t* rv = d_recv ? (staged ? d.rv : d_recv + 4 * lo) + 4 * pc[q] : nullptr;
            if ((r = child_call(kid, pc[q + 1] - pc[q], rp, st, rv, d.run)) != ZKV_OK) return r;
            HIP_TRY(hipSetDevice(kid->device));
        }
        if (staged) {
            HIP_TRY(hipMemcpyPeerAsync(d_status + lo, sdev, d.st, kid->device, m, d.run));
            if (d_recv) HIP_TRY(hipMemcpyPeerAsync(d_recv + 4 * lo, sdev, d.rv, kid->device, 4 * m, d.run));
        }
        HIP_TRY(hipEventRecord(d.ev_done, d.run));
        d.has_done = true;
        return ZKV_OK;
    };
    shard_parallel(c, used, [&](size_t k) { rc[k] = one(k); });
    for (int r : rc) if (r != ZKV_OK) return r;
    if (caller) {                                     // ... and the caller's stream continues after every shard has delivered its statuses
        HIP_TRY(hipSetDevice(sdev));
        for (size_t k = 0; k < used; k++) { size_t lo, hi; shard_range(n, used, k, &lo, &hi); if (hi > lo) HIP_TRY(hipStreamWaitEvent(caller, c->sh[k].ev_done, 0)); }
    }
    return ZKV_OK;
}
static void shards_free(zkv_ctx* c) {
    shard_workers_stop(c);
    for (size_t k = 0; k < c->sh.size(); k++) {
        zkv_ctx::ShardDev& d = c->sh[k];
        if (!d.run) continue;
        (void)hipSetDevice(c->shards[k]->device);
        (void)hipStreamSynchronize(d.run); (void)hipStreamSynchronize(d.copy);
        for (auto& r : d.row) { if (r) (void)hipFree(r); r = nullptr; }
        if (d.st) (void)hipFree(d.st);
        if (d.rv) (void)hipFree(d.rv);
        for (auto& e : d.ev_piece) if (e) (void)hipEventDestroy(e);
        if (d.ev_done) (void)hipEventDestroy(d.ev_done);
        (void)hipStreamDestroy(d.copy); (void)hipStreamDestroy(d.run);
        d = zkv_ctx::ShardDev();
    }
    for (auto& e : c->ev_in) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    for (auto& k : c->shards) { if (k) zkv_ctx_destroy(k); k = nullptr; }
    c->shards.clear(); c->sh.clear();
}
ZKV_EXPORT zkv_ctx* zkv_ctx_create_sharded(zkv_ctx* const* shards, size_t n_shards) {
    if (!shards || !n_shards || n_shards > 64) return nullptr;
    for (size_t k = 0; k < n_shards; k++) {
        const zkv_ctx* s = shards[k];
        if (!s || is_sharded(s) || s->vm != shards[0]->vm) return nullptr;
        for (size_t j = 0; j < k; j++) if (shards[j] == s) return nullptr;
        if (s->vm != ZKV_VM_RISC0 && s->vm != ZKV_VM_SP1 && s->vm != ZKV_VM_MIXED && s->vm != ZKV_VM_GROTH16 && s->vm != ZKV_VM_SP1_PLONK) return nullptr;
        // shards of one verifier: the same parameters everywhere (the host-visible state of shard 0 answers the getters)
        if (!s->initialized || memcmp(s->selector, shards[0]->selector, 4) || memcmp(s->control_id, shards[0]->control_id, 32) ||
            memcmp(s->gvk, shards[0]->gvk, sizeof s->gvk) || s->g_n_ic != shards[0]->g_n_ic || s->g_negate != shards[0]->g_negate ||
            memcmp(s->plonk_hash, shards[0]->plonk_hash, 32)) return nullptr;
        if (s->vm == ZKV_VM_SP1_PLONK && (memcmp(&s->pk_raw, &shards[0]->pk_raw, sizeof s->pk_raw) || memcmp(s->pk_g2, shards[0]->pk_g2, sizeof s->pk_g2))) return nullptr;
        if (s->vm == ZKV_VM_MIXED && memcmp(s->kid[0]->selector, shards[0]->kid[0]->selector, 4)) return nullptr;
    }
    zkv_ctx* c = new (std::nothrow) zkv_ctx();
    if (!c) return nullptr;
    const zkv_ctx* s0 = shards[0];
    c->vm = s0->vm; c->device = s0->device; c->initialized = s0->initialized; c->id_ge_r = s0->id_ge_r;
    memcpy(c->control_root_0, s0->control_root_0, 16); memcpy(c->control_root_1, s0->control_root_1, 16);
    memcpy(c->control_id, s0->control_id, 32); memcpy(c->selector, s0->selector, 4);
    c->consts = s0->consts; c->g_n_ic = s0->g_n_ic; c->g_negate = s0->g_negate; memcpy(c->gvk, s0->gvk, sizeof c->gvk);
    memcpy(c->plonk_hash, s0->plonk_hash, 32);
    c->shards.assign(shards, shards + n_shards);
    c->sh.resize(n_shards);
    return c;
}
ZKV_EXPORT size_t zkv_ctx_shard_count(const zkv_ctx* c) { return c ? c->shards.size() : 0; }
ZKV_EXPORT int zkv_ctx_shard_device(const zkv_ctx* c, size_t k) { return c && k < c->shards.size() ? c->shards[k]->device : ZKV_ERR_INVALID_ARG; }
// one shard per set bit of device_mask (bit d = HIP device d), lowest device first
template <class Mk> static zkv_ctx* create_multi(uint64_t device_mask, Mk mk) {
    std::vector<zkv_ctx*> kids;
    for (int d = 0; d < 64; d++) {
        if (!((device_mask >> d) & 1u)) continue;
        zkv_ctx* k = mk(d);
        if (!k) { for (auto* x : kids) zkv_ctx_destroy(x); return nullptr; }
        kids.push_back(k);
    }
    if (kids.empty()) return nullptr;
    zkv_ctx* c = zkv_ctx_create_sharded(kids.data(), kids.size());
    if (!c) for (auto* x : kids) zkv_ctx_destroy(x);
    return c;
}
ZKV_EXPORT zkv_ctx* zkv_risc0_ctx_create_multi(const uint8_t control_root[32], const uint8_t bn254_control_id[32], uint64_t device_mask) {
    if (!control_root || !bn254_control_id) return nullptr;
    return create_multi(device_mask, [&](int d) { return zkv_risc0_ctx_create(control_root, bn254_control_id, d); });
}
ZKV_EXPORT zkv_ctx* zkv_sp1_ctx_create_multi(uint64_t device_mask) { return create_multi(device_mask, [](int d) { return zkv_sp1_ctx_create(d); }); }
ZKV_EXPORT zkv_ctx* zkv_mixed_ctx_create_multi(const uint8_t control_root[32], const uint8_t bn254_control_id[32], uint64_t device_mask) {
    if (!control_root || !bn254_control_id) return nullptr;
    return create_multi(device_mask, [&](int d) { return zkv_mixed_ctx_create(control_root, bn254_control_id, d); });
}

// ------------------------------------------------------------------ mixed batches: per-proof VMType (common/types.rs:24-26)
ZKV_EXPORT zkv_ctx* zkv_mixed_ctx_create(const uint8_t control_root[32], const uint8_t bn254_control_id[32], int device) {
    if (!control_root || !bn254_control_id) return nullptr;
    zkv_ctx* c = new (std::nothrow) zkv_ctx();
    if (!c) return nullptr;
    c->vm = ZKV_VM_MIXED; c->device = device; c->initialized = true;
    memset(&c->consts, 0, sizeof c->consts);
    c->kid[0] = zkv_risc0_ctx_create(control_root, bn254_control_id, device);
    c->kid[1] = zkv_sp1_ctx_create(device);
    if (!c->kid[0] || !c->kid[1]) { zkv_ctx_destroy(c); return nullptr; }
    return c;
}
ZKV_EXPORT zkv_ctx* zkv_mixed_ctx_risc0(zkv_ctx* c) { if (is_sharded(c)) c = c->shards[0]; return c && c->vm == ZKV_VM_MIXED ? c->kid[0] : nullptr; }
ZKV_EXPORT zkv_ctx* zkv_mixed_ctx_sp1(zkv_ctx* c) { if (is_sharded(c)) c = c->shards[0]; return c && c->vm == ZKV_VM_MIXED ? c->kid[1] : nullptr; }

enum { MX_CNT = 0, MX_TOT, MX_POS, MX_IDX, MX_SEALS, MX_LEN, MX_A, MX_B, MX_PVOFF, MX_PVLEN, MX_ST, MX_RV,
       MX_H_VM, MX_H_SEALS, MX_H_SOFF, MX_H_A, MX_H_B, MX_H_BOFF, MX_H_ST, MX_H_RV };
// Everything device-resident; `seal_off` / `b_off` select the ragged layout, otherwise fixed strides.  Synchronises `s` once, after
// the partition, to learn the two sub-batch sizes.
static int run_mixed(zkv_ctx* c, size_t n, const uint8_t* d_vm, const uint8_t* d_seals, const uint64_t* d_seal_off, uint32_t seal_stride,
                     const uint8_t* d_a, const uint8_t* d_b, const uint64_t* d_b_off, uint32_t b_stride, uint32_t pv_len,
                     uint8_t* d_status, uint8_t* d_recv, hipStream_t s) {
    int rc;
    const size_t blocks = (n + 255) / 256;
    const size_t need[12] = {8 * blocks, 8, 4 * n, 4 * n, (size_t)ZKV_SEAL_BYTES * n, 4 * n, 32 * n, 32 * n, 8 * n, 4 * n, n, 4 * n};
    for (int k = 0; k < 12; k++) if ((rc = grow(&c->mx[k], &c->mx_cap[k], need[k])) != ZKV_OK) return rc;
    if ((rc = order_after_previous(c, s)) != ZKV_OK) return rc;
    MixedArgs a;
    memset(&a, 0, sizeof a);
    a.n = n; a.vm = d_vm; a.seals = d_seals; a.seal_off = d_seal_off; a.seal_stride = seal_stride;
    a.in_a = d_a; a.in_b = d_b; a.b_off = d_b_off; a.b_stride = b_stride; a.pv_len = pv_len;
    a.cnt = (uint32_t*)c->mx[MX_CNT]; a.totals = (uint32_t*)c->mx[MX_TOT]; a.pos = (uint32_t*)c->mx[MX_POS]; a.idx = (uint32_t*)c->mx[MX_IDX];
    a.c_seals = c->mx[MX_SEALS]; a.c_len = (uint32_t*)c->mx[MX_LEN]; a.c_a = c->mx[MX_A]; a.c_b = c->mx[MX_B];
    a.c_pvoff = (uint64_t*)c->mx[MX_PVOFF]; a.c_pvlen = (uint32_t*)c->mx[MX_PVLEN];
    a.status = d_status; a.recv = d_recv;
    launch_mixed_partition(a, (uint32_t*)c->mx[MX_CNT], (uint32_t*)c->mx[MX_TOT], s);
    HIP_TRY(hipGetLastError());
    uint32_t tot[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(tot, c->mx[MX_TOT], sizeof tot, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const size_t n0 = tot[0], n1 = tot[1];
    if (n0 + n1 > n) return ZKV_ERR_HIP;
    c->kid_ran[0] = n0 > 0; c->kid_ran[1] = n1 > 0;
    uint8_t *st = c->mx[MX_ST], *rv = c->mx[MX_RV];
    if ((rc = run_records(c->kid[0], n0, a.c_seals, a.c_len, a.c_a, a.c_b, nullptr, nullptr, nullptr, st, rv, s)) != ZKV_OK) return rc;
    if ((rc = run_records(c->kid[1], n1, a.c_seals + ZKV_SEAL_BYTES * n0, a.c_len + n0, a.c_a + 32 * n0, nullptr, d_b, a.c_pvoff + n0, a.c_pvlen + n0,
                          st + n0, rv + 4 * n0, s)) != ZKV_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    launch_mixed_return(n0 + n1, a.idx, st, rv, d_status, d_recv, s);
    HIP_TRY(hipGetLastError());
    return mark_done(c, s);
}
ZKV_EXPORT int zkv_mixed_verify_batch_dev(zkv_ctx* c, size_t n, const uint8_t* d_vm, const uint8_t* d_seals, const uint8_t* d_in_a, const uint8_t* d_in_b,
                                          size_t b_stride, size_t pv_len, uint8_t* d_status, uint8_t* d_recv, void* stream) {
    if (!c || c->vm != ZKV_VM_MIXED) return ZKV_ERR_WRONG_CTX;
    if (n && (!d_vm || !d_seals || !d_in_a || !d_in_b || !d_status || b_stride < 32 || pv_len > b_stride || b_stride > 0xFFFFFFFFu)) return ZKV_ERR_INVALID_ARG;
    if (!n) return ZKV_OK;
    if (n > 0xFFFFFFF0u) return ZKV_ERR_INVALID_ARG;
    if (is_sharded(c)) {
        const DevRow rows[4] = {{d_vm, 1}, {d_seals, ZKV_SEAL_BYTES}, {d_in_a, 32}, {d_in_b, b_stride}};
        return run_sharded_dev(c, n, rows, 4, d_status, d_recv, stream, [&](zkv_ctx* k, size_t m, const uint8_t* const* r, uint8_t* st, uint8_t* rv, hipStream_t s) {
            return zkv_mixed_verify_batch_dev(k, m, r[0], r[1], r[2], r[3], b_stride, pv_len, st, rv, s); });
    }
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_device_init(c);
    if (rc != ZKV_OK) return rc;
    return run_mixed(c, n, d_vm, d_seals, nullptr, ZKV_SEAL_BYTES, d_in_a, d_in_b, nullptr, (uint32_t)b_stride, (uint32_t)pv_len, d_status, d_recv,
                     stream ? (hipStream_t)stream : c->stream);
}
ZKV_EXPORT int zkv_mixed_verify_batch(zkv_ctx* c, size_t n, const uint8_t* vm, const uint8_t* seal_blob, const uint64_t* seal_off, const uint8_t* in_a,
                                      const uint8_t* in_b_blob, const uint64_t* in_b_off, uint8_t* status, uint8_t* recv) {
    if (!c || c->vm != ZKV_VM_MIXED) return ZKV_ERR_WRONG_CTX;
    if (n && (!vm || !seal_blob || !seal_off || !in_a || !in_b_blob || !in_b_off || !status)) return ZKV_ERR_INVALID_ARG;
    if (!n) return ZKV_OK;
    if (n > 0xFFFFFFF0u || !offsets_ok(seal_off, n) || !offsets_ok(in_b_off, n)) return ZKV_ERR_INVALID_ARG;
    for (size_t i = 0; i < n; i++)                   // journal_digest is a B256 in the reference (risc0/verifier.rs:82)
        if (vm[i] == ZKV_VM_RISC0 && in_b_off[i + 1] - in_b_off[i] != 32) return ZKV_ERR_INVALID_ARG;
    if (is_sharded(c))
        return run_sharded(c, n, [&](zkv_ctx* k, size_t lo, size_t hi) {
            return zkv_mixed_verify_batch(k, hi - lo, vm + lo, seal_blob, seal_off + lo, in_a + 32 * lo, in_b_blob, in_b_off + lo, status + lo, recv ? recv + 4 * lo : nullptr); });
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_device_init(c);
    if (rc != ZKV_OK) return rc;
    const uint64_t s0 = seal_off[0], sbytes = seal_off[n] - s0, b0 = in_b_off[0], bbytes = in_b_off[n] - b0;
    const size_t need[8] = {n, (size_t)sbytes + 8, 8 * (n + 1), 32 * n, (size_t)bbytes + 8, 8 * (n + 1), n, 4 * n};
    for (int k = 0; k < 8; k++) if ((rc = grow(&c->mx[MX_H_VM + k], &c->mx_cap[MX_H_VM + k], need[k])) != ZKV_OK) return rc;
    hipStream_t s = c->stream;
    if ((rc = order_after_previous(c, s)) != ZKV_OK) return rc;
    std::vector<uint64_t> so(seal_off, seal_off + n + 1), bo(in_b_off, in_b_off + n + 1);
    for (auto& v : so) v -= s0;
    for (auto& v : bo) v -= b0;
    HIP_TRY(hipMemcpyAsync(c->mx[MX_H_VM], vm, n, hipMemcpyHostToDevice, s));
    if (sbytes) HIP_TRY(hipMemcpyAsync(c->mx[MX_H_SEALS], seal_blob + s0, (size_t)sbytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->mx[MX_H_SOFF], so.data(), 8 * (n + 1), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->mx[MX_H_A], in_a, 32 * n, hipMemcpyHostToDevice, s));
    if (bbytes) HIP_TRY(hipMemcpyAsync(c->mx[MX_H_B], in_b_blob + b0, (size_t)bbytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->mx[MX_H_BOFF], bo.data(), 8 * (n + 1), hipMemcpyHostToDevice, s));
    if ((rc = run_mixed(c, n, c->mx[MX_H_VM], c->mx[MX_H_SEALS], (const uint64_t*)c->mx[MX_H_SOFF], 0, c->mx[MX_H_A], c->mx[MX_H_B],
                        (const uint64_t*)c->mx[MX_H_BOFF], 0, 0, c->mx[MX_H_ST], c->mx[MX_H_RV], s)) != ZKV_OK) return rc;
    HIP_TRY(hipMemcpyAsync(status, c->mx[MX_H_ST], n, hipMemcpyDeviceToHost, s));
    if (recv) HIP_TRY(hipMemcpyAsync(recv, c->mx[MX_H_RV], 4 * n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return ZKV_OK;
}

// ------------------------------------------------------------------ RISC Zero
ZKV_EXPORT zkv_ctx* zkv_risc0_ctx_new(int device) {
    zkv_ctx* c = new (std::nothrow) zkv_ctx();
    if (!c) return nullptr;
    c->vm = ZKV_VM_RISC0; c->device = device;
    host::risc0_consts(c->consts);
    return c;
}
ZKV_EXPORT int zkv_risc0_initialize(zkv_ctx* c, const uint8_t control_root[32], const uint8_t bn254_control_id[32], uint8_t* status) {
    if (!c || c->vm != ZKV_VM_RISC0 || !control_root || !bn254_control_id) return ZKV_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->initialized) { if (status) *status = ZKV_STATUS_ALREADY_INITIALIZED; return ZKV_OK; }   // verifier.rs:63-65
    host::split_digest(control_root, c->control_root_0, c->control_root_1);                        // :67-69
    memcpy(c->control_id, bn254_control_id, 32);                                                   // :70
    host::risc0_selector(control_root, bn254_control_id, c->selector);                             // :71-72
    uint32_t id[8];
    host::be_to_limbs(id, bn254_control_id);
    c->id_ge_r = !raw_lt_r(id);            // such a verifier fails every proof at groth16.rs:32
    c->initialized = true;
    if (status) *status = ZKV_STATUS_OK;
    return ZKV_OK;
}
ZKV_EXPORT zkv_ctx* zkv_risc0_ctx_create(const uint8_t control_root[32], const uint8_t bn254_control_id[32], int device) {
    zkv_ctx* c = zkv_risc0_ctx_new(device);
    if (!c) return nullptr;
    uint8_t st;
    if (zkv_risc0_initialize(c, control_root, bn254_control_id, &st) != ZKV_OK) { delete c; return nullptr; }
    return c;
}
ZKV_EXPORT void zkv_ctx_destroy(zkv_ctx* c) {
    if (!c) return;
    if (is_sharded(c)) shards_free(c);
    for (auto& k : c->kid) { if (k) zkv_ctx_destroy(k); k = nullptr; }
    ctx_free_device(c);
    delete c;
}
ZKV_EXPORT int zkv_risc0_get_selector(const zkv_ctx* c, uint8_t out[4]) {
    if (!c || c->vm != ZKV_VM_RISC0) return ZKV_ERR_WRONG_CTX;
    memcpy(out, c->selector, 4); return ZKV_OK;
}
ZKV_EXPORT int zkv_risc0_get_control_root(const zkv_ctx* c, uint8_t out_0[16], uint8_t out_1[16]) {
    if (!c || c->vm != ZKV_VM_RISC0) return ZKV_ERR_WRONG_CTX;
    memcpy(out_0, c->control_root_0, 16); memcpy(out_1, c->control_root_1, 16); return ZKV_OK;
}
ZKV_EXPORT int zkv_risc0_get_bn254_control_id(const zkv_ctx* c, uint8_t out[32]) {
    if (!c || c->vm != ZKV_VM_RISC0) return ZKV_ERR_WRONG_CTX;
    memcpy(out, c->control_id, 32); return ZKV_OK;
}
ZKV_EXPORT int zkv_risc0_get_verifier_key_digest(const zkv_ctx* c, uint8_t out[32]) {
    if (!c || c->vm != ZKV_VM_RISC0) return ZKV_ERR_WRONG_CTX;
    host::risc0_vk_digest(out); return ZKV_OK;
}
ZKV_EXPORT int zkv_risc0_is_initialized(const zkv_ctx* c) { return c && c->vm == ZKV_VM_RISC0 && c->initialized ? 1 : 0; }

ZKV_EXPORT int zkv_risc0_verify_batch(zkv_ctx* c, size_t n, const uint8_t* seal_blob, const uint64_t* seal_off, const uint8_t* image_ids,
                                      const uint8_t* journal_digests, uint8_t* status, uint8_t* recv) {
    if (!c || c->vm != ZKV_VM_RISC0) return ZKV_ERR_WRONG_CTX;
    if (n && (!image_ids || !journal_digests)) return ZKV_ERR_INVALID_ARG;
    if (is_sharded(c)) {
        if (n && (!seal_blob || !seal_off || !status)) return ZKV_ERR_INVALID_ARG;
        return run_sharded(c, n, [&](zkv_ctx* k, size_t lo, size_t hi) {
            return zkv_risc0_verify_batch(k, hi - lo, seal_blob, seal_off + lo, image_ids + 32 * lo, journal_digests + 32 * lo, status + lo, recv ? recv + 4 * lo : nullptr); });
    }
    return run_host_batch(c, n, seal_blob, seal_off, image_ids, journal_digests, nullptr, nullptr, status, recv);
}
ZKV_EXPORT int zkv_risc0_verify_integrity_batch(zkv_ctx* c, size_t n, const uint8_t* seal_blob, const uint64_t* seal_off,
                                                const uint8_t* claim_digests, uint8_t* status, uint8_t* recv) {
    if (!c || c->vm != ZKV_VM_RISC0) return ZKV_ERR_WRONG_CTX;
    if (n && !claim_digests) return ZKV_ERR_INVALID_ARG;
    if (is_sharded(c)) {
        if (n && (!seal_blob || !seal_off || !status)) return ZKV_ERR_INVALID_ARG;
        return run_sharded(c, n, [&](zkv_ctx* k, size_t lo, size_t hi) {
            return zkv_risc0_verify_integrity_batch(k, hi - lo, seal_blob, seal_off + lo, claim_digests + 32 * lo, status + lo, recv ? recv + 4 * lo : nullptr); });
    }
    return run_host_batch(c, n, seal_blob, seal_off, claim_digests, nullptr, nullptr, nullptr, status, recv);
}
ZKV_EXPORT int zkv_risc0_verify(zkv_ctx* c, const uint8_t* seal, size_t seal_len, const uint8_t image_id[32], const uint8_t journal_digest[32],
                                uint8_t* status, uint8_t recv[4]) {
    uint64_t off[2] = {0, seal_len};
    uint8_t dummy = 0;
    if (!seal && seal_len) return ZKV_ERR_INVALID_ARG;
    return zkv_risc0_verify_batch(c, 1, seal ? seal : &dummy, off, image_id, journal_digest, status, recv);
}
ZKV_EXPORT int zkv_risc0_verify_integrity(zkv_ctx* c, const uint8_t* seal, size_t seal_len, const uint8_t claim_digest[32], uint8_t* status,
                                          uint8_t recv[4]) {
    uint64_t off[2] = {0, seal_len};
    uint8_t dummy = 0;
    if (!seal && seal_len) return ZKV_ERR_INVALID_ARG;
    return zkv_risc0_verify_integrity_batch(c, 1, seal ? seal : &dummy, off, claim_digest, status, recv);
}
ZKV_EXPORT int zkv_risc0_verify_batch_dev(zkv_ctx* c, size_t n, const uint8_t* d_seals, const uint8_t* d_image_ids,
                                          const uint8_t* d_journal_digests, uint8_t* d_status, uint8_t* d_recv, void* stream) {
    if (!c || c->vm != ZKV_VM_RISC0) return ZKV_ERR_WRONG_CTX;
    if (n && !d_journal_digests) return ZKV_ERR_INVALID_ARG;
    if (is_sharded(c)) {
        if (n && (!d_seals || !d_image_ids || !d_status)) return ZKV_ERR_INVALID_ARG;
        const DevRow rows[3] = {{d_seals, ZKV_SEAL_BYTES}, {d_image_ids, 32}, {d_journal_digests, 32}};
        return run_sharded_dev(c, n, rows, 3, d_status, d_recv, stream, [&](zkv_ctx* k, size_t m, const uint8_t* const* r, uint8_t* st, uint8_t* rv, hipStream_t s) {
            return zkv_risc0_verify_batch_dev(k, m, r[0], r[1], r[2], st, rv, s); });
    }
    return run_dev_batch(c, n, d_seals, d_image_ids, d_journal_digests, nullptr, 0, d_status, d_recv, stream);
}

// ------------------------------------------------------------------ RISC Zero verifier sets (many instances, one VK)
ZKV_EXPORT zkv_ctx* zkv_risc0_set_create(size_t n_instances, const uint8_t* control_roots, const uint8_t* bn254_control_ids, int device) {
    if (!n_instances || n_instances > ((size_t)1 << 24) || !control_roots || !bn254_control_ids) return nullptr;
    zkv_ctx* c = new (std::nothrow) zkv_ctx();
    if (!c) return nullptr;
    c->vm = ZKV_VM_RISC0_SET; c->device = device; c->initialized = true;
    host::risc0_consts(c->consts);
    c->inst_raw.resize(n_instances);
    for (size_t i = 0; i < n_instances; i++) {
        memcpy(c->inst_raw[i].control_root, control_roots + 32 * i, 32);
        memcpy(c->inst_raw[i].control_id, bn254_control_ids + 32 * i, 32);
    }
    return c;
}
ZKV_EXPORT size_t zkv_risc0_set_size(const zkv_ctx* c) { return c && c->vm == ZKV_VM_RISC0_SET ? c->inst_raw.size() : 0; }
ZKV_EXPORT int zkv_risc0_set_get_selector(zkv_ctx* c, size_t instance, uint8_t out[4]) {
    if (!c || c->vm != ZKV_VM_RISC0_SET) return ZKV_ERR_WRONG_CTX;
    if (!out || instance >= c->inst_raw.size()) return ZKV_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_device_init(c);                        // selectors are derived by the set-up kernel
    if (rc != ZKV_OK) return rc;
    const uint32_t s = c->inst_host[instance].selector_be;
    out[0] = (uint8_t)(s >> 24); out[1] = (uint8_t)(s >> 16); out[2] = (uint8_t)(s >> 8); out[3] = (uint8_t)s;
    return ZKV_OK;
}
// shared driver: host pointers when `dev` is false (one chunk at a time, synchronous), device pointers otherwise (asynchronous)
static int run_set_batch(zkv_ctx* c, size_t n, const uint32_t* inst, const uint8_t* blob, const uint64_t* off, const uint8_t* ids, const uint8_t* jds,
                         uint8_t* status, uint8_t* recv, bool dev, void* stream) {
    if (!c || c->vm != ZKV_VM_RISC0_SET) return ZKV_ERR_WRONG_CTX;
    if (n && (!inst || !blob || (!dev && !off) || !ids || !jds || !status)) return ZKV_ERR_INVALID_ARG;
    if (n && !dev && !offsets_ok(off, n)) return ZKV_ERR_INVALID_ARG;
    if (!n) return ZKV_OK;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_ready(c, n);
    if (rc != ZKV_OK) return rc;
    hipStream_t s = dev && stream ? (hipStream_t)stream : c->stream;
    if ((rc = order_after_previous(c, s)) != ZKV_OK) return rc;
    const size_t cap = c->ws.cap;
    std::vector<uint64_t> rel(dev ? 0 : cap + 1);
    for (size_t base = 0; base < n; base += cap) {
        const size_t m = n - base < cap ? n - base : cap;
        PrepArgs a;
        memset(&a, 0, sizeof a);
        a.n = m; a.inst_tab = c->d_inst; a.n_inst = (uint32_t)c->inst_raw.size();
        if (dev) {
            a.blob = blob + base * ZKV_SEAL_BYTES; a.stride = ZKV_SEAL_BYTES; a.inst = inst + base;
            a.in32_a = ids + 32 * base; a.in32_b = jds + 32 * base; a.status = status + base; a.recv = recv ? recv + 4 * base : nullptr;
        } else {
            const uint64_t b0 = off[base], bytes = off[base + m] - b0;
            if ((rc = grow(&c->d_blob, &c->blob_cap, (size_t)bytes + 8)) != ZKV_OK) return rc;
            for (size_t i = 0; i <= m; i++) rel[i] = off[base + i] - b0;
            HIP_TRY(hipMemcpyAsync(c->d_off, rel.data(), sizeof(uint64_t) * (m + 1), hipMemcpyHostToDevice, s));
            if (bytes) HIP_TRY(hipMemcpyAsync(c->d_blob, blob + b0, (size_t)bytes, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(c->d_inst_idx, inst + base, sizeof(uint32_t) * m, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(c->d_a, ids + 32 * base, 32 * m, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(c->d_b, jds + 32 * base, 32 * m, hipMemcpyHostToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));               // rel[] is reused by the next chunk
            a.blob = c->d_blob; a.off = c->d_off; a.inst = c->d_inst_idx; a.in32_a = c->d_a; a.in32_b = c->d_b;
            a.status = c->d_status; a.recv = c->d_recv;
        }
        enqueue_chunk(c, a, s, base + cap >= n);
        HIP_TRY(hipGetLastError());
        if (!dev) {
            HIP_TRY(hipMemcpyAsync(status + base, c->d_status, m, hipMemcpyDeviceToHost, s));
            if (recv) HIP_TRY(hipMemcpyAsync(recv + 4 * base, c->d_recv, 4 * m, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
    }
    return dev ? mark_done(c, s) : ZKV_OK;
}
ZKV_EXPORT int zkv_risc0_set_verify_batch(zkv_ctx* c, size_t n, const uint32_t* instance, const uint8_t* seal_blob, const uint64_t* seal_off,
                                          const uint8_t* image_ids, const uint8_t* journal_digests, uint8_t* status, uint8_t* recv) {
    if (recv && n) memset(recv, 0, 4 * n);
    return run_set_batch(c, n, instance, seal_blob, seal_off, image_ids, journal_digests, status, recv, false, nullptr);
}
ZKV_EXPORT int zkv_risc0_set_verify_batch_dev(zkv_ctx* c, size_t n, const uint32_t* d_instance, const uint8_t* d_seals, const uint8_t* d_image_ids,
                                              const uint8_t* d_journal_digests, uint8_t* d_status, uint8_t* d_recv, void* stream) {
    return run_set_batch(c, n, d_instance, d_seals, nullptr, d_image_ids, d_journal_digests, d_status, d_recv, true, stream);
}
// compute_vk_x for (instance, claim halves): the per-instance signals come from the device table
ZKV_EXPORT int zkv_risc0_set_vk_x_batch(zkv_ctx* c, size_t n, const uint32_t* instance, const uint8_t* var_signals, uint8_t* out) {
    if (!c || c->vm != ZKV_VM_RISC0_SET) return ZKV_ERR_WRONG_CTX;
    if (n && (!instance || !var_signals || !out)) return ZKV_ERR_INVALID_ARG;
    if (!n) return ZKV_OK;
    for (size_t i = 0; i < n; i++) if (instance[i] >= c->inst_raw.size()) return ZKV_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_ready(c, n);
    if (rc != ZKV_OK) return rc;
    if ((rc = order_after_previous(c, c->stream)) != ZKV_OK) return rc;
    const size_t cap = c->ws.cap;
    for (size_t base = 0; base < n; base += cap) {
        size_t m = n - base < cap ? n - base : cap;
        if ((rc = grow(&c->d_blob, &c->blob_cap, m * 64 + 8)) != ZKV_OK) return rc;
        if ((rc = grow(&c->d_pv, &c->pv_cap, m * 64 + 8)) != ZKV_OK) return rc;
        HIP_TRY(hipMemcpyAsync(c->d_blob, var_signals + 64 * base, 64 * m, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_inst_idx, instance + base, sizeof(uint32_t) * m, hipMemcpyHostToDevice, c->stream));
        launch_vk_x(m, c->d_tab, c->m16, c->d_inst, c->d_inst_idx, c->d_blob, c->d_pv, c->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out + 64 * base, c->d_pv, 64 * m, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return ZKV_OK;
}

// ------------------------------------------------------------------ SP1
ZKV_EXPORT zkv_ctx* zkv_sp1_ctx_create(int device) {
    zkv_ctx* c = new (std::nothrow) zkv_ctx();
    if (!c) return nullptr;
    c->vm = ZKV_VM_SP1; c->device = device; c->initialized = true;
    memset(&c->consts, 0, sizeof c->consts);
    return c;
}
ZKV_EXPORT int zkv_sp1_verifier_hash(uint8_t out[32]) { memcpy(out, host::SP1_VERIFIER_HASH, 32); return ZKV_OK; }
ZKV_EXPORT const char* zkv_sp1_version(void) { return host::SP1_VERSION; }
ZKV_EXPORT int zkv_sp1_verify_batch(zkv_ctx* c, size_t n, const uint8_t* vkeys, const uint8_t* pv_blob, const uint64_t* pv_off,
                                    const uint8_t* proof_blob, const uint64_t* proof_off, uint8_t* status, uint8_t* recv) {
    if (!c || c->vm != ZKV_VM_SP1) return ZKV_ERR_WRONG_CTX;
    if (n && (!vkeys || !pv_blob || !pv_off)) return ZKV_ERR_INVALID_ARG;
    if (is_sharded(c)) {
        if (n && (!proof_blob || !proof_off || !status)) return ZKV_ERR_INVALID_ARG;
        return run_sharded(c, n, [&](zkv_ctx* k, size_t lo, size_t hi) {
            return zkv_sp1_verify_batch(k, hi - lo, vkeys + 32 * lo, pv_blob, pv_off + lo, proof_blob, proof_off + lo, status + lo, recv ? recv + 4 * lo : nullptr); });
    }
    return run_host_batch(c, n, proof_blob, proof_off, vkeys, nullptr, pv_blob, pv_off, status, recv);
}
ZKV_EXPORT int zkv_sp1_verify_proof(zkv_ctx* c, const uint8_t vkey[32], const uint8_t* pv, size_t pv_len, const uint8_t* proof, size_t proof_len,
                                    uint8_t* status, uint8_t recv[4]) {
    uint64_t poff[2] = {0, proof_len}, voff[2] = {0, pv_len};
    uint8_t dummy = 0;
    if ((!pv && pv_len) || (!proof && proof_len)) return ZKV_ERR_INVALID_ARG;
    return zkv_sp1_verify_batch(c, 1, vkey, pv ? pv : &dummy, voff, proof ? proof : &dummy, poff, status, recv);
}
ZKV_EXPORT int zkv_sp1_verify_batch_dev(zkv_ctx* c, size_t n, const uint8_t* d_vkeys, const uint8_t* d_pv, size_t pv_len, const uint8_t* d_proofs,
                                        uint8_t* d_status, uint8_t* d_recv, void* stream) {
    if (!c || c->vm != ZKV_VM_SP1) return ZKV_ERR_WRONG_CTX;
    if (n && !d_pv) return ZKV_ERR_INVALID_ARG;
    if (is_sharded(c)) {
        if (n && (!d_proofs || !d_vkeys || !d_status)) return ZKV_ERR_INVALID_ARG;
        const DevRow rows[3] = {{d_vkeys, 32}, {d_pv, pv_len}, {d_proofs, ZKV_SEAL_BYTES}};
        return run_sharded_dev(c, n, rows, 3, d_status, d_recv, stream, [&](zkv_ctx* k, size_t m, const uint8_t* const* r, uint8_t* st, uint8_t* rv, hipStream_t s) {
            return zkv_sp1_verify_batch_dev(k, m, r[0], r[1], pv_len, r[2], st, rv, s); });
    }
    return run_dev_batch(c, n, d_proofs, d_vkeys, nullptr, d_pv, pv_len, d_status, d_recv, stream);
}

// ------------------------------------------------------------------ SP1 PLONK (no reference code: parity unpinned)
ZKV_EXPORT zkv_ctx* zkv_sp1_plonk_ctx_create(const uint8_t* vk, size_t vk_len, const uint8_t verifier_hash[32], int device) {
    if (!vk || !verifier_hash || vk_len < 7 * 32) return nullptr;
    uint32_t w[7][8];
    for (int k = 0; k < 7; k++) host::be_to_limbs(w[k], vk + 32 * k);
    auto small = [](const uint32_t* x) { for (int i = 1; i < 8; i++) if (x[i]) return false; return true; };
    // SP1's circuit: two public inputs and exactly one BSB22 commitment (the 27-word proof layout); other shapes are not supported
    if (!small(w[4]) || !small(w[5]) || !small(w[6]) || w[5][0] != 1 || w[4][0] != 2) return nullptr;
    const size_t n_c = w[5][0];
    if (vk_len != 7 * 32 + (8 + n_c) * 64 + 256) return nullptr;
    zkv_ctx* c = new (std::nothrow) zkv_ctx();
    if (!c) return nullptr;
    c->vm = ZKV_VM_SP1_PLONK; c->device = device; c->initialized = true;
    memset(&c->consts, 0, sizeof c->consts);
    memset(&c->pk_raw, 0, sizeof c->pk_raw);
    memcpy(c->pk_raw.size, w[0], 32); memcpy(c->pk_raw.size_inv, w[1], 32); memcpy(c->pk_raw.gen, w[2], 32); memcpy(c->pk_raw.coset, w[3], 32);
    c->pk_raw.nb_public = w[4][0]; c->pk_raw.n_c = w[5][0]; c->pk_raw.cci = w[6][0];
    for (size_t p = 0; p < 8 + n_c; p++) {
        host::be_to_limbs(c->pk_raw.pts[p][0], vk + 224 + 64 * p); host::be_to_limbs(c->pk_raw.pts[p][1], vk + 256 + 64 * p);
    }
    memcpy(c->pk_g2, vk + 224 + 64 * (8 + n_c), 256);
    memcpy(c->plonk_hash, verifier_hash, 32);
    return c;
}
ZKV_EXPORT int zkv_sp1_plonk_verifier_hash(const zkv_ctx* c, uint8_t out[32]) {
    if (!c || c->vm != ZKV_VM_SP1_PLONK) return ZKV_ERR_WRONG_CTX;
    memcpy(out, c->plonk_hash, 32); return ZKV_OK;
}
ZKV_EXPORT int zkv_sp1_plonk_verify_batch(zkv_ctx* c, size_t n, const uint8_t* vkeys, const uint8_t* pv_blob, const uint64_t* pv_off,
                                          const uint8_t* proof_blob, const uint64_t* proof_off, uint8_t* status, uint8_t* recv) {
    if (!c || c->vm != ZKV_VM_SP1_PLONK) return ZKV_ERR_WRONG_CTX;
    if (n && (!vkeys || !pv_blob || !pv_off)) return ZKV_ERR_INVALID_ARG;
    if (is_sharded(c)) {
        if (n && (!proof_blob || !proof_off || !status)) return ZKV_ERR_INVALID_ARG;
        return run_sharded(c, n, [&](zkv_ctx* k, size_t lo, size_t hi) {
            return zkv_sp1_plonk_verify_batch(k, hi - lo, vkeys + 32 * lo, pv_blob, pv_off + lo, proof_blob, proof_off + lo, status + lo, recv ? recv + 4 * lo : nullptr); });
    }
    return run_host_batch(c, n, proof_blob, proof_off, vkeys, nullptr, pv_blob, pv_off, status, recv);
}
ZKV_EXPORT int zkv_sp1_plonk_verify_proof(zkv_ctx* c, const uint8_t vkey[32], const uint8_t* pv, size_t pv_len, const uint8_t* proof, size_t proof_len,
                                          uint8_t* status, uint8_t recv[4]) {
    uint64_t poff[2] = {0, proof_len}, voff[2] = {0, pv_len};
    uint8_t dummy = 0;
    if ((!pv && pv_len) || (!proof && proof_len)) return ZKV_ERR_INVALID_ARG;
    return zkv_sp1_plonk_verify_batch(c, 1, vkey, pv ? pv : &dummy, voff, proof ? proof : &dummy, poff, status, recv);
}
ZKV_EXPORT int zkv_sp1_plonk_verify_batch_dev(zkv_ctx* c, size_t n, const uint8_t* d_vkeys, const uint8_t* d_pv, size_t pv_len, const uint8_t* d_proofs,
                                              uint8_t* d_status, uint8_t* d_recv, void* stream) {
    if (!c || c->vm != ZKV_VM_SP1_PLONK) return ZKV_ERR_WRONG_CTX;
    if (n && !d_pv) return ZKV_ERR_INVALID_ARG;
    if (is_sharded(c)) {
        if (n && (!d_proofs || !d_vkeys || !d_status)) return ZKV_ERR_INVALID_ARG;
        const DevRow rows[3] = {{d_vkeys, 32}, {d_pv, pv_len}, {d_proofs, ZKV_PLONK_PROOF_BYTES}};
        return run_sharded_dev(c, n, rows, 3, d_status, d_recv, stream, [&](zkv_ctx* k, size_t m, const uint8_t* const* r, uint8_t* st, uint8_t* rv, hipStream_t s) {
            return zkv_sp1_plonk_verify_batch_dev(k, m, r[0], r[1], pv_len, r[2], st, rv, s); });
    }
    return run_dev_batch(c, n, d_proofs, d_vkeys, nullptr, d_pv, pv_len, d_status, d_recv, stream);
}

// ------------------------------------------------------------------ on-chain wire layer (eth_call batches)
ZKV_EXPORT int zkv_abi_function_selector(const char* signature, uint8_t out[4]) {
    if (!signature || !out) return ZKV_ERR_INVALID_ARG;
    host::fn_selector(signature, out);
    return ZKV_OK;
}
ZKV_EXPORT size_t zkv_risc0_encode_verify_call(const uint8_t* seal, size_t seal_len, const uint8_t image_id[32], const uint8_t journal_digest[32],
                                               uint8_t* out, size_t cap) {
    const size_t need = 4 + 96 + 32 * (seal_len + 1);
    if (!out || cap < need || (seal_len && !seal) || !image_id || !journal_digest) return need;
    memcpy(out, host::selectors().risc0[host::R0_VERIFY], 4);
    host::abi_word_u32(out + 4, 0x60); memcpy(out + 36, image_id, 32); memcpy(out + 68, journal_digest, 32);
    host::abi_u8_array(out + 100, seal, seal_len);
    return need;
}
ZKV_EXPORT size_t zkv_risc0_encode_verify_integrity_call(const uint8_t* seal, size_t seal_len, const uint8_t claim_digest[32], uint8_t* out, size_t cap) {
    const size_t need = 4 + 64 + 32 * (seal_len + 1);
    if (!out || cap < need || (seal_len && !seal) || !claim_digest) return need;
    memcpy(out, host::selectors().risc0[host::R0_VERIFY_INTEGRITY], 4);
    host::abi_word_u32(out + 4, 0x40); memcpy(out + 36, claim_digest, 32);
    host::abi_u8_array(out + 68, seal, seal_len);
    return need;
}
ZKV_EXPORT size_t zkv_sp1_encode_verify_proof_call(const uint8_t program_vkey[32], const uint8_t* pv, size_t pv_len, const uint8_t* proof, size_t proof_len,
                                                   uint8_t* out, size_t cap) {
    const size_t need = 4 + 96 + 32 * (pv_len + 1) + 32 * (proof_len + 1);
    if (!out || cap < need || (pv_len && !pv) || (proof_len && !proof) || !program_vkey) return need;
    memcpy(out, host::selectors().sp1[host::SP1_VERIFY_PROOF], 4);
    memcpy(out + 4, program_vkey, 32); host::abi_word_u32(out + 36, 0x60); host::abi_word_u32(out + 68, 0x80 + 32 * (uint64_t)pv_len);
    size_t k = host::abi_u8_array(out + 100, pv, pv_len);
    host::abi_u8_array(out + 100 + k, proof, proof_len);
    return need;
}

// Return / revert data of a verify-class call from its status (success: `true` word for RISC Zero, nothing for SP1).
static void verify_returndata(const zkv_ctx* c, uint8_t st, const uint8_t* recv, uint8_t* out, uint32_t* out_len, uint8_t* reverted) {
    if (st == ZKV_STATUS_OK) {
        *reverted = 0;
        if (c->vm == ZKV_VM_RISC0) { host::abi_word_u32(out, 1); *out_len = 32; } else *out_len = 0;
        return;
    }
    *reverted = 1;
    if (st == ZKV_STATUS_BAD_CALLDATA) { *out_len = 0; return; }
    int k = zkv_status_abi_encode(c->vm, st, recv, c->vm == ZKV_VM_RISC0 ? c->selector : host::SP1_VERIFIER_HASH, out);
    *out_len = k > 0 ? (uint32_t)k : 0;
}
// Calls the device left as BAD_CALLDATA: either one of the constant-size methods (answered here) or really undecodable.
static void host_method(const zkv_ctx* c, const uint8_t* cd, size_t len, uint8_t* out, uint32_t* out_len, uint8_t* reverted) {
    *out_len = 0; *reverted = 1;
    if (len < 4) return;
    const host::Selectors& S = host::selectors();
    if (c->vm == ZKV_VM_RISC0) {
        int k = -1;
        for (int i = 0; i < host::R0_COUNT; i++) if (!memcmp(S.risc0[i], cd, 4)) k = i;
        if (k < 0 || k == host::R0_VERIFY || k == host::R0_VERIFY_INTEGRITY) return;
        if (k == host::R0_INITIALIZE) {                      // eth_call simulates the transaction; nothing is stored
            if (len != 4 + 64) return;
            if (c->initialized) *out_len = (uint32_t)zkv_status_abi_encode(ZKV_VM_RISC0, ZKV_STATUS_ALREADY_INITIALIZED, nullptr, nullptr, out);
            else *reverted = 0;
            return;
        }
        if (len != 4) return;
        *reverted = 0; *out_len = 32;
        if (k == host::R0_IS_INITIALIZED) host::abi_word_u32(out, c->initialized ? 1 : 0);
        else if (k == host::R0_GET_SELECTOR) host::abi_word_left(out, c->selector, 4);
        else if (k == host::R0_GET_CONTROL_ROOT) { host::abi_word_left(out, c->control_root_0, 16); host::abi_word_left(out + 32, c->control_root_1, 16); *out_len = 64; }
        else if (k == host::R0_GET_BN254_CONTROL_ID) memcpy(out, c->control_id, 32);
        else host::risc0_vk_digest(out);
    } else {
        int k = -1;
        for (int i = 0; i < host::SP1_COUNT; i++) if (!memcmp(S.sp1[i], cd, 4)) k = i;
        if (k <= host::SP1_VERIFY_PROOF || len != 4) return;
        *reverted = 0;
        if (k == host::SP1_FN_VERIFIER_HASH) { memcpy(out, host::SP1_VERIFIER_HASH, 32); *out_len = 32; return; }
        const size_t vl = strlen(host::SP1_VERSION);
        host::abi_word_u32(out, 0x20); host::abi_word_u32(out + 32, vl); memset(out + 64, 0, 32); memcpy(out + 64, host::SP1_VERSION, vl);
        *out_len = 96;
    }
}

// Decode + verify one chunk whose calldata and offsets are already on the device.
static int enqueue_wire_chunk(zkv_ctx* c, size_t m, const uint8_t* d_cd, const uint64_t* d_cdoff, uint64_t cd_bytes, uint8_t* d_status, uint8_t* d_recv,
                              hipStream_t s, bool timed, hipEvent_t decoded = nullptr) {
    int rc;
    if ((rc = grow(&c->d_blob, &c->blob_cap, m * ZKV_SEAL_BYTES + 8)) != ZKV_OK) return rc;
    if (c->vm == ZKV_VM_SP1 && (rc = grow(&c->d_pv, &c->pv_cap, (size_t)(cd_bytes / 32) + 64)) != ZKV_OK) return rc;
    const host::Selectors& S = host::selectors();
    WireArgs w;
    memset(&w, 0, sizeof w);
    w.n = m; w.cd = d_cd; w.off = d_cdoff; w.cd_bytes = cd_bytes;
    w.seals = c->d_blob; w.seal_len = c->d_len; w.in_a = c->d_a; w.in_b = c->d_b; w.kind = c->d_kind;
    w.pv = c->d_pv; w.pv_off = c->d_pvoff; w.pv_len = c->d_pvlen;
    if (timed) (void)hipEventRecord(c->ev_wire[0], s);
    if (c->vm == ZKV_VM_RISC0) {
        w.sel_a_be = be32_of(S.risc0[host::R0_VERIFY]); w.sel_b_be = be32_of(S.risc0[host::R0_VERIFY_INTEGRITY]);
        launch_wire_risc0(w, s);
    } else {
        w.sel_a_be = be32_of(S.sp1[host::SP1_VERIFY_PROOF]);
        launch_wire_sp1(w, s);
    }
    if (timed) { (void)hipEventRecord(c->ev_wire[1], s); c->wire_timed = true; }
    if (decoded) (void)hipEventRecord(decoded, s);           // the calldata buffer may be overwritten from here on
    PrepArgs a;
    memset(&a, 0, sizeof a);
    a.n = m; a.blob = c->d_blob; a.off = nullptr; a.stride = ZKV_SEAL_BYTES; a.len = c->d_len;
    a.in32_a = c->d_a;
    if (c->vm == ZKV_VM_RISC0) {
        a.in32_b = c->d_b; a.kind = c->d_kind;
        a.selector_be = be32_of(c->selector);
        a.force_fail = c->id_ge_r ? 1u : 0u;
        a.not_initialized = c->initialized ? 0u : 1u;
    } else {
        a.pv_blob = c->d_pv; a.pv_off = c->d_pvoff; a.pv_len = c->d_pvlen;
        a.selector_be = be32_of(host::SP1_VERIFIER_HASH);
    }
    a.status = d_status; a.recv = d_recv;
    enqueue_chunk(c, a, s, timed);
    return ZKV_OK;
}
static int wire_buffers(zkv_ctx*) { return ZKV_OK; }          // the copy stream and its events are created with the context

// Host calldata: 8-12 KB per proof cross PCIe, as much time as the verification itself.  Chunks of at most 2^16 requests
// (enough lanes to fill the chip) are double-buffered: the H2D copy of chunk k+1 runs on its own stream while chunk k is
// decoded and verified; statuses stay on the device until the whole batch is done, so the host never waits inside the loop.
static size_t wire_host_chunk(size_t cap) {
    const char* e = getenv("ZKV_WIRE_HOST_CHUNK");
    size_t v = e ? (size_t)strtoull(e, nullptr, 10) : (size_t)1 << 16;
    if (v < 1) v = 1;
    return v < cap ? v : cap;
}
static int run_eth_call_batch(zkv_ctx* c, size_t n, const uint8_t* blob, const uint64_t* off, uint8_t* reverted, uint8_t* returndata,
                              uint32_t* returndata_len, uint8_t* status) {
    if (!c || (n && (!blob || !off || !reverted || !returndata || !returndata_len))) return ZKV_ERR_INVALID_ARG;
    if (!n) return ZKV_OK;
    if (!offsets_ok(off, n)) return ZKV_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_ready(c, n);
    if (rc != ZKV_OK) return rc;
    if ((rc = wire_buffers(c)) != ZKV_OK) return rc;
    if ((rc = order_after_previous(c, c->stream)) != ZKV_OK) return rc;
    const size_t chunk = wire_host_chunk(c->ws.cap);
    if ((rc = grow(&c->d_st_all, &c->st_all_cap, n)) != ZKV_OK || (rc = grow(&c->d_rv_all, &c->rv_all_cap, 4 * n)) != ZKV_OK) return rc;
    uint64_t max_bytes = 0;
    for (size_t base = 0; base < n; base += chunk) {
        size_t m = n - base < chunk ? n - base : chunk;
        uint64_t bytes = off[base + m] - off[base];
        if (bytes > max_bytes) max_bytes = bytes;
    }
    // all device buffers are sized before the loop: growing one frees it, which synchronises the device
    for (int b = 0; b < 2; b++)
        if ((n > chunk || b == 0) && (rc = grow(&c->d_cd[b], &c->cd_cap[b], (size_t)max_bytes + 8)) != ZKV_OK) return rc;
    if ((rc = grow(&c->d_blob, &c->blob_cap, chunk * ZKV_SEAL_BYTES + 8)) != ZKV_OK) return rc;
    if (c->vm == ZKV_VM_SP1 && (rc = grow(&c->d_pv, &c->pv_cap, (size_t)(max_bytes / 32) + 64)) != ZKV_OK) return rc;
    std::vector<uint64_t> rel[2];
    size_t k = 0;
    for (size_t base = 0; base < n; base += chunk, k++) {
        const size_t m = n - base < chunk ? n - base : chunk;
        const int b = (int)(k & 1);
        const uint64_t b0 = off[base], bytes = off[base + m] - b0;
        if (k >= 2) HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->ev_decoded[b], 0));     // chunk k-2 has been decoded out of this buffer
        HIP_TRY(hipStreamSynchronize(c->copy_stream));                                    // rel[b] of chunk k-2 is no longer being read
        rel[b].resize(m + 1);
        for (size_t i = 0; i <= m; i++) rel[b][i] = off[base + i] - b0;
        HIP_TRY(hipMemcpyAsync(c->d_cdoff[b], rel[b].data(), sizeof(uint64_t) * (m + 1), hipMemcpyHostToDevice, c->copy_stream));
        if (bytes) HIP_TRY(hipMemcpyAsync(c->d_cd[b], blob + b0, (size_t)bytes, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(hipEventRecord(c->ev_copied[b], c->copy_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_copied[b], 0));
        if ((rc = enqueue_wire_chunk(c, m, c->d_cd[b], c->d_cdoff[b], bytes, c->d_st_all + base, c->d_rv_all + 4 * base, c->stream,
                                     base + chunk >= n, c->ev_decoded[b])) != ZKV_OK) return rc;
        HIP_TRY(hipGetLastError());
    }
    std::vector<uint8_t> st(n), rv(4 * n);
    HIP_TRY(hipMemcpyAsync(st.data(), c->d_st_all, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(rv.data(), c->d_rv_all, 4 * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->copy_stream));
    for (size_t i = 0; i < n; i++) {
        uint8_t* out = returndata + i * ZKV_RETURNDATA_STRIDE;
        if (st[i] == ZKV_STATUS_BAD_CALLDATA) host_method(c, blob + off[i], (size_t)(off[i + 1] - off[i]), out, &returndata_len[i], &reverted[i]);
        else verify_returndata(c, st[i], rv.data() + 4 * i, out, &returndata_len[i], &reverted[i]);
        if (status) status[i] = st[i];
    }
    return ZKV_OK;
}
ZKV_EXPORT int zkv_risc0_eth_call_batch(zkv_ctx* c, size_t n, const uint8_t* calldata_blob, const uint64_t* calldata_off, uint8_t* reverted,
                                        uint8_t* returndata, uint32_t* returndata_len, uint8_t* status) {
    if (!c || c->vm != ZKV_VM_RISC0) return ZKV_ERR_WRONG_CTX;
    if (is_sharded(c)) {
        if (n && (!calldata_blob || !calldata_off || !reverted || !returndata || !returndata_len)) return ZKV_ERR_INVALID_ARG;
        return run_sharded(c, n, [&](zkv_ctx* k, size_t lo, size_t hi) {
            return zkv_risc0_eth_call_batch(k, hi - lo, calldata_blob, calldata_off + lo, reverted + lo, returndata + lo * ZKV_RETURNDATA_STRIDE, returndata_len + lo, status ? status + lo : nullptr); });
    }
    return run_eth_call_batch(c, n, calldata_blob, calldata_off, reverted, returndata, returndata_len, status);
}
ZKV_EXPORT int zkv_sp1_eth_call_batch(zkv_ctx* c, size_t n, const uint8_t* calldata_blob, const uint64_t* calldata_off, uint8_t* reverted,
                                      uint8_t* returndata, uint32_t* returndata_len, uint8_t* status) {
    if (!c || c->vm != ZKV_VM_SP1) return ZKV_ERR_WRONG_CTX;
    if (is_sharded(c)) {
        if (n && (!calldata_blob || !calldata_off || !reverted || !returndata || !returndata_len)) return ZKV_ERR_INVALID_ARG;
        return run_sharded(c, n, [&](zkv_ctx* k, size_t lo, size_t hi) {
            return zkv_sp1_eth_call_batch(k, hi - lo, calldata_blob, calldata_off + lo, reverted + lo, returndata + lo * ZKV_RETURNDATA_STRIDE, returndata_len + lo, status ? status + lo : nullptr); });
    }
    return run_eth_call_batch(c, n, calldata_blob, calldata_off, reverted, returndata, returndata_len, status);
}
// Device-resident calldata: verify-class calls only, statuses stay on the device.
ZKV_EXPORT int zkv_eth_call_batch_dev(zkv_ctx* c, size_t n, const uint8_t* d_calldata, const uint64_t* d_calldata_off, uint64_t calldata_bytes,
                                      uint8_t* d_status, uint8_t* d_recv, void* stream) {
    if (is_sharded(c)) c = c->shards[0];                 // calldata offsets are absolute into one blob: this entry point stays on one GPU
    if (!c || (c->vm != ZKV_VM_RISC0 && c->vm != ZKV_VM_SP1)) return ZKV_ERR_WRONG_CTX;
    if (n && (!d_calldata || !d_calldata_off || !d_status)) return ZKV_ERR_INVALID_ARG;
    if (!n) return ZKV_OK;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_ready(c, n);
    if (rc != ZKV_OK) return rc;
    if ((rc = wire_buffers(c)) != ZKV_OK) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const size_t cap = c->ws.cap;
    if ((rc = order_after_previous(c, s)) != ZKV_OK) return rc;
    for (size_t base = 0; base < n; base += cap) {
        size_t m = n - base < cap ? n - base : cap;
        // offsets are absolute into d_calldata, so chunks share the blob pointer; the public-values scratch is sized for the whole blob
        if ((rc = enqueue_wire_chunk(c, m, d_calldata, d_calldata_off + base, calldata_bytes, d_status + base, d_recv ? d_recv + 4 * base : nullptr, s,
                                     base + cap >= n)) != ZKV_OK) return rc;
    }
    HIP_TRY(hipGetLastError());
    return mark_done(c, s);
}
ZKV_EXPORT int zkv_eth_call_returndata(const zkv_ctx* c, uint8_t status, const uint8_t recv_selector[4], uint8_t out[ZKV_RETURNDATA_STRIDE],
                                       uint32_t* out_len, uint8_t* reverted) {
    static const uint8_t zero[4] = {0, 0, 0, 0};
    if (!c || !out || !out_len || !reverted || (c->vm != ZKV_VM_RISC0 && c->vm != ZKV_VM_SP1) || status > ZKV_STATUS_BAD_CALLDATA) return ZKV_ERR_INVALID_ARG;
    verify_returndata(c, status, recv_selector ? recv_selector : zero, out, out_len, reverted);
    return ZKV_OK;
}
ZKV_EXPORT int zkv_ctx_last_wire_ms(zkv_ctx* c, float* out_ms) {
    if (is_sharded(c)) c = c->shards[0];
    if (!c || !out_ms) return ZKV_ERR_INVALID_ARG;
    if (!c->dev_ready || !c->wire_timed) return ZKV_ERR_NO_DEVICE;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev_wire[1]));
    HIP_TRY(hipEventElapsedTime(out_ms, c->ev_wire[0], c->ev_wire[1]));
    return ZKV_OK;
}

// ------------------------------------------------------------------ precompile-level batches
ZKV_EXPORT zkv_ctx* zkv_bn254_ctx_create(int device) {
    zkv_ctx* c = new (std::nothrow) zkv_ctx();
    if (!c) return nullptr;
    c->vm = ZKV_VM_BN254; c->device = device; c->initialized = true;
    memset(&c->consts, 0, sizeof c->consts);
    return c;
}
// kind 0 = ecAdd (128 -> 64), 1 = ecMul (96 -> 64), 2 = ecPairing (k*192 -> result byte)
static int run_precompile(zkv_ctx* c, int kind, size_t n, size_t k, const uint8_t* in, uint8_t* out, uint8_t* ok) {
    if (!c || c->vm != ZKV_VM_BN254) return ZKV_ERR_WRONG_CTX;
    if (n && (!in || !out || !ok)) return ZKV_ERR_INVALID_ARG;
    if (!n) return ZKV_OK;
    std::lock_guard<std::mutex> lk(c->mu);
    const bool resident = kind == 2 && pairing_group((uint32_t)k) && n > dual_below();
    int rc = ctx_ready(c, resident ? n * k : n);
    if (rc != ZKV_OK) return rc;
    const size_t in_sz = kind == 0 ? 128 : kind == 1 ? 96 : 192 * k, out_sz = kind == 2 ? 1 : 64;
    // ecPairing calls of 2 .. 8 pairs keep all their pairs resident (one Miller loop per call, launch_pairing): k workspace slots per call
    const size_t cap = resident ? c->ws.cap / k : c->ws.cap;
    if ((rc = order_after_previous(c, c->stream)) != ZKV_OK) return rc;
    for (size_t base = 0; base < n; base += cap) {
        size_t m = n - base < cap ? n - base : cap;
        if ((rc = grow(&c->d_blob, &c->blob_cap, m * in_sz + 8)) != ZKV_OK) return rc;
        if ((rc = grow(&c->d_pv, &c->pv_cap, m * out_sz + 8)) != ZKV_OK) return rc;
        if (in_sz) HIP_TRY(hipMemcpyAsync(c->d_blob, in + base * in_sz, m * in_sz, hipMemcpyHostToDevice, c->stream));
        if (kind == 0) launch_ecadd(m, c->d_blob, c->d_pv, c->d_status, c->stream);
        else if (kind == 1) launch_ecmul(m, c->d_blob, c->d_pv, c->d_status, c->stream);
        else if (m <= dual_below()) launch_pairing_w(m, (uint32_t)k, c->d_blob, c->ws, c->d_pv, c->d_status, c->stream);      // few calls: latency
        else launch_pairing(m, (uint32_t)k, c->d_blob, c->ws, c->d_pv, c->d_status, c->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out + base * out_sz, c->d_pv, m * out_sz, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(ok + base, c->d_status, m, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return ZKV_OK;
}
ZKV_EXPORT int zkv_bn254_ecadd_batch(zkv_ctx* c, size_t n, const uint8_t* in, uint8_t* out, uint8_t* ok) { return run_precompile(c, 0, n, 0, in, out, ok); }
ZKV_EXPORT int zkv_bn254_ecmul_batch(zkv_ctx* c, size_t n, const uint8_t* in, uint8_t* out, uint8_t* ok) { return run_precompile(c, 1, n, 0, in, out, ok); }
// The ecPairing seam with calldata, results and verdicts resident in HBM: enqueued on the caller's stream (or the context's), no copy,
// no synchronisation.
ZKV_EXPORT int zkv_bn254_pairing_batch_dev(zkv_ctx* c, size_t n, size_t k, const uint8_t* d_in, uint8_t* d_result, uint8_t* d_ok, void* stream) {
    if (!c || c->vm != ZKV_VM_BN254) return ZKV_ERR_WRONG_CTX;
    if (k > 64 || (n && (!d_result || !d_ok || (k && !d_in)))) return ZKV_ERR_INVALID_ARG;
    if (!n) return ZKV_OK;
    std::lock_guard<std::mutex> lk(c->mu);
    const bool resident = pairing_group((uint32_t)k) && n > dual_below();
    int rc = ctx_ready(c, resident ? n * k : n);
    if (rc != ZKV_OK) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if ((rc = order_after_previous(c, s)) != ZKV_OK) return rc;
    const size_t cap = resident ? c->ws.cap / k : c->ws.cap;
    for (size_t base = 0; base < n; base += cap) {
        const size_t m = n - base < cap ? n - base : cap;
        const uint8_t* in = k ? d_in + base * 192 * k : d_in;
        if (m <= dual_below()) launch_pairing_w(m, (uint32_t)k, in, c->ws, d_result + base, d_ok + base, s);
        else launch_pairing(m, (uint32_t)k, in, c->ws, d_result + base, d_ok + base, s);
        HIP_TRY(hipGetLastError());
    }
    return mark_done(c, s);
}
ZKV_EXPORT int zkv_bn254_pairing_batch(zkv_ctx* c, size_t n, size_t k, const uint8_t* in, uint8_t* result, uint8_t* ok) {
    if (k > 64) return ZKV_ERR_INVALID_ARG;
    uint8_t dummy = 0;
    return run_precompile(c, 2, n, k, k ? in : &dummy, result, ok);
}

// ------------------------------------------------------------------ Groth16 core, arbitrary verification key
ZKV_EXPORT zkv_ctx* zkv_groth16_ctx_create(const uint8_t* vk_words, size_t n_ic, int vm_type, int device) {
    if (!vk_words || n_ic < 1 || n_ic > MAX_IC || (vm_type != ZKV_VM_RISC0 && vm_type != ZKV_VM_SP1)) return nullptr;
    zkv_ctx* c = new (std::nothrow) zkv_ctx();
    if (!c) return nullptr;
    c->vm = ZKV_VM_GROTH16; c->device = device; c->initialized = true;
    memset(&c->consts, 0, sizeof c->consts);
    memcpy(c->gvk, vk_words, 448 + 64 * n_ic);
    c->g_n_ic = (uint32_t)n_ic; c->g_negate = vm_type == ZKV_VM_RISC0;
    return c;
}
ZKV_EXPORT int zkv_groth16_verify_batch(zkv_ctx* c, size_t n, const uint8_t* proofs, const uint8_t* signals, uint8_t* verified) {
    if (!c || c->vm != ZKV_VM_GROTH16) return ZKV_ERR_WRONG_CTX;
    const uint32_t n_sig = c->g_n_ic - 1;
    if (n && (!proofs || !verified || (n_sig && !signals))) return ZKV_ERR_INVALID_ARG;
    if (!n) return ZKV_OK;
    if (is_sharded(c))
        return run_sharded(c, n, [&](zkv_ctx* k, size_t lo, size_t hi) {
            return zkv_groth16_verify_batch(k, hi - lo, proofs + 256 * lo, n_sig ? signals + (size_t)32 * n_sig * lo : signals, verified + lo); });
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_ready(c, n);
    if (rc != ZKV_OK) return rc;
    const size_t cap = c->ws.cap;
    if ((rc = order_after_previous(c, c->stream)) != ZKV_OK) return rc;
    for (size_t base = 0; base < n; base += cap) {
        size_t m = n - base < cap ? n - base : cap;
        if ((rc = grow(&c->d_blob, &c->blob_cap, m * 256 + 8)) != ZKV_OK) return rc;
        if ((rc = grow(&c->d_pv, &c->pv_cap, m * 32 * n_sig + 8)) != ZKV_OK) return rc;
        HIP_TRY(hipMemcpyAsync(c->d_blob, proofs + 256 * base, 256 * m, hipMemcpyHostToDevice, c->stream));
        if (n_sig) HIP_TRY(hipMemcpyAsync(c->d_pv, signals + (size_t)32 * n_sig * base, (size_t)32 * n_sig * m, hipMemcpyHostToDevice, c->stream));
        PrepArgs a;
        memset(&a, 0, sizeof a);
        a.n = m; a.blob = c->d_blob; a.in32_a = c->d_pv; a.n_sig = n_sig; a.negate_a = c->g_negate ? 1u : 0u;
        a.force_fail = c->vk_invalid ? 1u : 0u;
        a.status = c->d_status; a.recv = nullptr;
        enqueue_chunk(c, a, c->stream, true);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(verified + base, c->d_status, m, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < m; i++) verified[base + i] = verified[base + i] == ZKV_STATUS_OK ? 1 : 0;
    }
    return ZKV_OK;
}

// ------------------------------------------------------------------ Groth16 core pieces
ZKV_EXPORT int zkv_ctx_vk_x_batch(zkv_ctx* c, size_t n, const uint8_t* var_signals, uint8_t* out) {
    if (is_sharded(c)) c = c->shards[0];
    if (!c || c->vm == ZKV_VM_BN254 || c->vm == ZKV_VM_RISC0_SET || c->vm == ZKV_VM_MIXED || c->vm == ZKV_VM_SP1_PLONK) return ZKV_ERR_WRONG_CTX;
    if (c->vm == ZKV_VM_RISC0 && !c->initialized) return ZKV_ERR_INVALID_ARG;
    if (n && (!var_signals || !out)) return ZKV_ERR_INVALID_ARG;
    if (!n) return ZKV_OK;
    // per-proof signals: two for the RISC Zero / SP1 keys, all n_ic - 1 for a generic key (k_vk_x reads n_var x 32 bytes per proof)
    const size_t sig = 32 * (size_t)(c->vm == ZKV_VM_GROTH16 ? c->g_n_ic - 1 : 2);
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_ready(c, n);
    if (rc != ZKV_OK) return rc;
    const size_t cap = c->ws.cap;
    if ((rc = order_after_previous(c, c->stream)) != ZKV_OK) return rc;
    for (size_t base = 0; base < n; base += cap) {
        size_t m = n - base < cap ? n - base : cap;
        if ((rc = grow(&c->d_blob, &c->blob_cap, m * sig + 8)) != ZKV_OK) return rc;
        if ((rc = grow(&c->d_pv, &c->pv_cap, m * 64 + 8)) != ZKV_OK) return rc;
        if (sig) HIP_TRY(hipMemcpyAsync(c->d_blob, var_signals + sig * base, sig * m, hipMemcpyHostToDevice, c->stream));
        launch_vk_x(m, c->d_tab, c->m16, nullptr, nullptr, c->d_blob, c->d_pv, c->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out + 64 * base, c->d_pv, 64 * m, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return ZKV_OK;
}

// ------------------------------------------------------------------ diagnostics: multiplication-rate and issue-rate microbenchmarks
// shared driver of the two microbenchmarks: `issue` selects k_diag_issue (64 instructions per loop trip and lane) over k_diag_mulmod
// (four primitive calls per trip; kind 1 counts two multiplications per call)
static int run_diag(int device, bool issue, int kind, int waves_per_simd, uint32_t iters, double* per_s, double* shader_clock_ghz) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return ZKV_ERR_NO_DEVICE; }
    if (device < 0 || device >= n || !device_is_gfx950(device)) return ZKV_ERR_NO_DEVICE;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    const unsigned blocks = (unsigned)p.multiProcessorCount * 4u * (unsigned)waves_per_simd;      // one 64-lane workgroup per wave slot
    uint32_t* d_out = nullptr; unsigned long long* d_clk = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = ZKV_ERR_HIP;
    float ms = 0;
    unsigned long long clk[2] = {0, 0};
    auto launch = [&](uint32_t it) { if (issue) launch_diag_issue(kind, blocks, it, d_out, d_clk, nullptr); else launch_diag_mulmod(kind, blocks, it, d_out, d_clk, nullptr); };
    if (hipMalloc(&d_out, (size_t)blocks * 64 * 4) != hipSuccess || hipMalloc(&d_clk, 16) != hipSuccess) goto done;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) goto done;
    launch(iters / 8 + 1);                                                                 // warm-up (code fetch, clocks)
    if (hipDeviceSynchronize() != hipSuccess) goto done;
    if (hipEventRecord(e0, nullptr) != hipSuccess) goto done;
    launch(iters);
    if (hipEventRecord(e1, nullptr) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) goto done;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost) != hipSuccess) goto done;
    *per_s = (double)blocks * 64.0 * (issue ? 64.0 : 4.0 * (kind == 1 ? 2.0 : 1.0)) * (double)iters / ((double)ms * 1e-3);
    if (shader_clock_ghz) *shader_clock_ghz = clk[1] ? 0.1 * (double)clk[0] / (double)clk[1] : 0.0;     // s_memrealtime ticks at 100 MHz
    rc = ZKV_OK;
done:
    if (rc != ZKV_OK) (void)hipGetLastError();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d_out) (void)hipFree(d_out);
    if (d_clk) (void)hipFree(d_clk);
    return rc;
}
ZKV_EXPORT int zkv_diag_mulmod_rate(int device, int kind, int waves_per_simd, uint32_t iters, double* mulmods_per_s, double* shader_clock_ghz) {
    if (kind < 0 || kind > 4 || waves_per_simd < 1 || waves_per_simd > 8 || !iters || !mulmods_per_s) return ZKV_ERR_INVALID_ARG;
    return run_diag(device, false, kind, waves_per_simd, iters, mulmods_per_s, shader_clock_ghz);
}
ZKV_EXPORT int zkv_diag_issue_rate(int device, int kind, int waves_per_simd, uint32_t iters, double* lane_instr_per_s, double* shader_clock_ghz) {
    if (kind < 0 || kind > 2 || waves_per_simd < 1 || waves_per_simd > 8 || !iters || !lane_instr_per_s) return ZKV_ERR_INVALID_ARG;
    return run_diag(device, true, kind, waves_per_simd, iters, lane_instr_per_s, shader_clock_ghz);
}

// ------------------------------------------------------------------ shared
ZKV_EXPORT int zkv_ctx_vm(const zkv_ctx* c) { return c ? c->vm : ZKV_ERR_INVALID_ARG; }
ZKV_EXPORT int zkv_ctx_set_lanes_per_proof(zkv_ctx* c, int lanes) {
    if (!c || (lanes != 0 && lanes != 2 && lanes != 16 && lanes != 64 && lanes != 128)) return ZKV_ERR_INVALID_ARG;
    if (is_sharded(c)) { for (auto* k : c->shards) { const int rc = zkv_ctx_set_lanes_per_proof(k, lanes); if (rc != ZKV_OK) return rc; } return ZKV_OK; }
    if (c->vm == ZKV_VM_MIXED) {                         // the two verifiers behind the tag run the stages
        for (auto* k : c->kid) { const int rc = zkv_ctx_set_lanes_per_proof(k, lanes); if (rc != ZKV_OK) return rc; }
    }
    std::lock_guard<std::mutex> lk(c->mu);
    c->lanes = lanes;
    return ZKV_OK;
}
// Aggregate check on / off (zkv_agg.h).  seed32 = nullptr draws the 32 secret bytes from the operating system.
ZKV_EXPORT int zkv_ctx_set_aggregate_check(zkv_ctx* c, int enable, const uint8_t* seed32) {
    if (!c || (enable != 0 && enable != 1 && enable != 16 && enable != 32 && enable != 64 && enable != 128 && enable != 256)) return ZKV_ERR_INVALID_ARG;
    uint8_t seed[32];
    if (enable) {
        if (seed32) memcpy(seed, seed32, 32);
        else if (getrandom(seed, 32, 0) != 32) return ZKV_ERR_INVALID_ARG;
    }
    if (is_sharded(c)) {                                 // every shard its own seed, derived from this one (or drawn afresh)
        for (size_t k = 0; k < c->shards.size(); k++) {
            uint8_t sk[32];
            if (enable && seed32) { uint8_t buf[36]; memcpy(buf, seed, 32); buf[32] = (uint8_t)(k >> 24); buf[33] = (uint8_t)(k >> 16); buf[34] = (uint8_t)(k >> 8); buf[35] = (uint8_t)k; host::sha256_host(buf, 36, sk); }
            const int rc = zkv_ctx_set_aggregate_check(c->shards[k], enable, enable && seed32 ? sk : nullptr);
            { volatile uint8_t* w = sk; for (int i = 0; i < 32; i++) w[i] = 0; }
            if (rc != ZKV_OK) return rc;
        }
        return ZKV_OK;
    }
    if (c->vm == ZKV_VM_MIXED) {
        for (int k = 0; k < 2; k++) {
            uint8_t sk[32];
            if (enable && seed32) { uint8_t buf[33]; memcpy(buf, seed, 32); buf[32] = (uint8_t)k; host::sha256_host(buf, 33, sk); }
            const int rc = zkv_ctx_set_aggregate_check(c->kid[k], enable, enable && seed32 ? sk : nullptr);
            { volatile uint8_t* w = sk; for (int i = 0; i < 32; i++) w[i] = 0; }
            if (rc != ZKV_OK) return rc;
        }
        return ZKV_OK;
    }
    if (c->vm != ZKV_VM_RISC0 && c->vm != ZKV_VM_RISC0_SET && c->vm != ZKV_VM_SP1 && c->vm != ZKV_VM_GROTH16 && c->vm != ZKV_VM_SP1_PLONK)
        return enable ? ZKV_ERR_INVALID_ARG : ZKV_OK;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->agg_on = enable != 0;
        if (enable) {
            c->agg_auto = enable == 1; c->agg_sub = enable == 1 ? 32u : (uint32_t)enable;
            c->agg_resnap = c->dev_ready;                       // counters of earlier runs (other sizes) are not this setting's evidence; a fresh context starts from zero
            if (!c->dev_ready) c->agg_seen[0] = c->agg_seen[1] = 0;
            c->agg_pause = c->agg_pause_len = 0;
            c->agg_os_seed = seed32 == nullptr; c->agg_key_age = 0;
            for (int i = 0; i < 8; i++) c->agg_seed.w[i] = be32_of(seed + 4 * i);
        }
    }
    volatile uint8_t* wipe = seed;                              // the secret does not stay on this stack
    for (int i = 0; i < 32; i++) wipe[i] = 0;
    return ZKV_OK;
}
// {sub-batches checked, sub-batches that failed and were verified proof by proof} since the context was set up; the calling thread
// must have synchronised with the batches it wants counted.
ZKV_EXPORT int zkv_ctx_aggregate_counters(zkv_ctx* c, uint64_t out[2]) {
    if (!c || !out) return ZKV_ERR_INVALID_ARG;
    out[0] = out[1] = 0;
    if (is_sharded(c) || c->vm == ZKV_VM_MIXED) {
        const size_t nk = is_sharded(c) ? c->shards.size() : 2;
        for (size_t k = 0; k < nk; k++) {
            uint64_t o[2];
            const int rc = zkv_ctx_aggregate_counters(is_sharded(c) ? c->shards[k] : c->kid[k], o);
            if (rc != ZKV_OK) return rc;
            out[0] += o[0]; out[1] += o[1];
        }
        return ZKV_OK;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->dev_ready || !c->d_agg_cnt) return ZKV_OK;
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long v[2];
    HIP_TRY(hipMemcpy(v, c->d_agg_cnt, sizeof v, hipMemcpyDeviceToHost));
    out[0] = v[0]; out[1] = v[1];
    return ZKV_OK;
}
ZKV_EXPORT int zkv_diag_wait_faults(int device, uint64_t* out) {
    if (!out) return ZKV_ERR_INVALID_ARG;
    *out = 0;
    if (device < 0 || device >= zkv_device_count()) return ZKV_ERR_NO_DEVICE;
    HIP_TRY(hipSetDevice(device));
    unsigned long long v = 0;
    if (zkv::read_wait_faults(&v) != 0) return ZKV_ERR_HIP;
    *out = v;
    return ZKV_OK;
}
ZKV_EXPORT int zkv_ctx_shard_peer_access(zkv_ctx* c, size_t shard) {
    if (!c || !is_sharded(c) || shard >= c->sh.size()) return ZKV_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    return c->sh[shard].peer;
}
ZKV_EXPORT int zkv_host_register(void* ptr, size_t bytes) {
    if (!ptr || !bytes) return ZKV_ERR_INVALID_ARG;
    if (zkv_device_count() < 1) return ZKV_ERR_NO_DEVICE;
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterPortable));
    return ZKV_OK;
}
ZKV_EXPORT int zkv_host_unregister(void* ptr) {
    if (!ptr) return ZKV_ERR_INVALID_ARG;
    if (zkv_device_count() < 1) return ZKV_ERR_NO_DEVICE;
    HIP_TRY(hipHostUnregister(ptr));
    return ZKV_OK;
}
ZKV_EXPORT int zkv_ctx_reserve(zkv_ctx* c, size_t n) {
    if (!c) return ZKV_ERR_INVALID_ARG;
    if (is_sharded(c)) {                                 // every shard for its share of an n-proof batch
        const size_t used = shards_used(c, n ? n : 1);
        for (size_t k = 0; k < used; k++) { size_t lo, hi; shard_range(n ? n : 1, used, k, &lo, &hi); const int rc = zkv_ctx_reserve(c->shards[k], hi - lo); if (rc != ZKV_OK) return rc; }
        return ZKV_OK;
    }
    if (c->vm == ZKV_VM_MIXED) {                         // either VM may own the whole batch
        int rc = zkv_ctx_reserve(c->kid[0], n);
        if (rc == ZKV_OK) rc = zkv_ctx_reserve(c->kid[1], n);
        if (rc != ZKV_OK) return rc;
        std::lock_guard<std::mutex> lk(c->mu);
        return ctx_device_init(c);
    }
    std::lock_guard<std::mutex> lk(c->mu);
    return ctx_ready(c, n);
}
ZKV_EXPORT int zkv_ctx_synchronize(zkv_ctx* c) {
    if (!c) return ZKV_ERR_INVALID_ARG;
    if (is_sharded(c)) {                                 // every shard's device, and the shards' own streams (status copies back to the source GPU)
        for (size_t k = 0; k < c->shards.size(); k++) {
            const int rc = zkv_ctx_synchronize(c->shards[k]);
            if (rc != ZKV_OK) return rc;
            if (c->sh[k].run) { HIP_TRY(hipSetDevice(c->shards[k]->device)); HIP_TRY(hipStreamSynchronize(c->sh[k].run)); }
        }
        return ZKV_OK;
    }
    // a mixed context has work in flight as soon as ANY of its three contexts is set up (an all-SP1 batch never touches the RISC Zero child)
    const bool any = c->dev_ready || (c->vm == ZKV_VM_MIXED && ((c->kid[0] && c->kid[0]->dev_ready) || (c->kid[1] && c->kid[1]->dev_ready)));
    if (!any) return ZKV_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    return ZKV_OK;
}
ZKV_EXPORT int zkv_ctx_last_stage_ms(zkv_ctx* c, float out_ms[5]) {
    if (!c || !out_ms) return ZKV_ERR_INVALID_ARG;
    if (is_sharded(c)) return zkv_ctx_last_stage_ms(c->shards[0], out_ms);          // shards run side by side: shard 0 stands for all
    if (c->vm == ZKV_VM_MIXED) {
        // the two sub-batches run one after the other: stage times add up.  Only the children that ran in the MOST RECENT mixed call
        // count (an unused child is either not set up or still holds the event times of an earlier batch).
        for (int i = 0; i < 5; i++) out_ms[i] = 0.0f;
        for (int k = 0; k < 2; k++) {
            if (!c->kid_ran[k]) continue;
            float a[5];
            const int rc = zkv_ctx_last_stage_ms(c->kid[k], a);
            if (rc != ZKV_OK) return rc;
            for (int i = 0; i < 5; i++) out_ms[i] += a[i];
        }
        return ZKV_OK;
    }
    if (!c->dev_ready) return ZKV_ERR_NO_DEVICE;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev[5]));
    for (int i = 0; i < 5; i++) HIP_TRY(hipEventElapsedTime(&out_ms[i], c->ev[i], c->ev[i + 1]));
    return ZKV_OK;
}
ZKV_EXPORT int zkv_status_abi_encode(int vm, uint8_t status, const uint8_t received[4], const uint8_t expected[4], uint8_t out[68]) {
    // keccak-256 selectors of the reference's Solidity custom errors
    static const uint8_t sel[5][4] = {{0, 0, 0, 0}, {0x43, 0x9c, 0xc0, 0xcd}, {0xf9, 0x2e, 0xe8, 0xa9}, {0x0d, 0xc1, 0x49, 0xf0}, {0xe3, 0xe9, 0x43, 0x26}};
    static const uint8_t mism[2][4] = {{0xb8, 0xb3, 0x8d, 0x4c}, {0x98, 0x80, 0x66, 0xa1}};
    if (vm == ZKV_VM_SP1_PLONK) vm = ZKV_VM_SP1;           // same ISp1Verifier errors
    if (!out || (vm != ZKV_VM_RISC0 && vm != ZKV_VM_SP1)) return ZKV_ERR_INVALID_ARG;
    if (status == ZKV_STATUS_OK) return 0;
    if (status == ZKV_STATUS_SELECTOR_MISMATCH) {
        if (!received || !expected) return ZKV_ERR_INVALID_ARG;
        memset(out, 0, 68);
        memcpy(out, mism[vm], 4); memcpy(out + 4, received, 4); memcpy(out + 36, expected, 4);
        return 68;
    }
    if (status > ZKV_STATUS_SELECTOR_MISMATCH) return ZKV_ERR_INVALID_ARG;
    memcpy(out, sel[status], 4);
    return 4;
}
