// Internal interface between the C-ABI translation unit and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "zkv_verify.h"

namespace zkv {

// Per-chunk workspace in HBM, struct-of-arrays: word k of proof i lives at base[k * cap + i], so the 64 lanes
// of a wavefront read/write 256 contiguous bytes per word (coalesced).
constexpr int WS_PREP_WORDS = 64 + 8 * MAX_VAR;   // ax ay cx cy (4x8) | bx.c0 bx.c1 by.c0 by.c1 (4x8) | per-proof scalars (MAX_VAR x 8)
constexpr int WS_NORM_WORDS = 48;   // axs ays lxs lys cxs cys
constexpr int WS_F_WORDS = 96;      // Fp12 Miller value (slot F of the final exponentiation)
constexpr int WS_FE_WORDS = 7 * 96; // cold Fp12 slots of the final exponentiation: E, Y1, Y3, Y4 and the window slots x^3, x^5, x^7
struct Workspace {
    uint32_t* prep; uint32_t* norm; uint32_t* f; uint32_t* fe; uint32_t* flags;
    uint32_t* g2bad;            // 1 = B failed the subgroup check (own word: the check may run beside the MSM, which owns `flags`)
    size_t cap;
};

constexpr int ZKV_BLOCK = 64;       // one wavefront per workgroup: one proof per lane, no cross-lane traffic

__device__ __forceinline__ Fp ws_ld(const uint32_t* base, size_t cap, int word0, size_t i) {
    Fp r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = base[(size_t)(word0 + k) * cap + i];
    return r;
}
__device__ __forceinline__ void ws_st(uint32_t* base, size_t cap, int word0, size_t i, const Fp& a) {
#pragma unroll
    for (int k = 0; k < 8; k++) base[(size_t)(word0 + k) * cap + i] = a.v[k];
}

struct PrepArgs {
    size_t n;
    const uint8_t* blob;        // seals / proofs
    const uint64_t* off;        // n+1 offsets, or nullptr for fixed stride
    uint32_t stride;            // bytes per record when off == nullptr
    const uint8_t* in32_a;      // risc0: image_ids (or claim digests when in32_b == nullptr); sp1: program vkeys
    const uint8_t* in32_b;      // risc0: journal digests; sp1: unused
    const uint8_t* pv_blob;     // sp1 public values
    const uint64_t* pv_off;     // sp1: n+1 offsets or nullptr
    uint32_t pv_stride;
    uint32_t selector_be;       // expected selector as big-endian word
    uint32_t force_fail;        // context-level VerificationFailed (risc0 bn254_control_id >= R, invalid generic VK)
    uint32_t n_sig;             // generic Groth16 batches: signals per proof (in32_a holds n x n_sig x 32 bytes)
    uint32_t negate_a;          // generic Groth16 batches: VMType::Risc0 negates A
    // wire-layer batches (k_wire.hip decodes calldata into fixed-stride records): per-proof decoded seal length
    // (0xFFFFFFFF = undecodable calldata), per-proof method (1 = verifyIntegrity: in32_a is the claim digest) and
    // per-proof public-values length (pv_off then holds n start offsets).
    const uint32_t* len;
    const uint8_t* kind;
    const uint32_t* pv_len;
    // verifier sets (zkv_risc0_set_*): per-proof instance index; selector and context-level failure come from the table
    const uint32_t* inst; const InstTab* inst_tab; uint32_t n_inst;
    uint32_t* plonk_tab;        // PLONK batches: PLONK_TAB_WORDS words per proof for the per-proof window tables of the MSMs (zkv_plonk.h)
    uint32_t not_initialized;   // wire-layer batches on an un-initialised RISC Zero verifier: decodable calls get InvalidInitialization
    uint8_t* status; uint8_t* recv;
};

// eth_call calldata decode (k_wire.hip): one wavefront per request.
struct WireArgs {
    size_t n;
    const uint8_t* cd;          // calldata blob
    const uint64_t* off;        // n+1 offsets
    uint64_t cd_bytes;          // size of the blob: a request whose offsets leave [0, cd_bytes] or run backwards is never read
    uint32_t sel_a_be, sel_b_be;    // risc0: verify / verifyIntegrity; sp1: verifyProof / unused
    uint8_t* seals;             // n x 260: first min(L, 260) decoded seal / proof bytes
    uint32_t* seal_len;         // n: decoded length L, or 0xFFFFFFFF when the calldata is not a canonical verify call
    uint8_t* in_a; uint8_t* in_b;   // n x 32 each: risc0 image id (claim digest) / journal digest; sp1 program vkey / unused
    uint8_t* kind;              // n: risc0 0 = verify, 1 = verifyIntegrity
    uint8_t* pv; uint64_t* pv_off; uint32_t* pv_len;    // sp1 public values: pv_off[i] = off[i] / 32
};
void launch_wire_risc0(const WireArgs& a, hipStream_t s);
void launch_wire_sp1(const WireArgs& a, hipStream_t s);

void launch_setup(const VkRaw* d_raw, VkTables* d_tab, hipStream_t s);
void launch_prep_risc0(const PrepArgs& a, const Risc0Consts& k, const Workspace& ws, hipStream_t s);
void launch_prep_sp1(const PrepArgs& a, const Workspace& ws, hipStream_t s);
void launch_prep_groth16(const PrepArgs& a, const Workspace& ws, hipStream_t s);
void launch_setup_msm16(const VkTables* d_tab, const Msm16& m, G1A* tab, uint32_t rows, hipStream_t s);
void launch_msm(size_t n, const VkTables* d_tab, const Msm16& m16, const InstTab* inst_tab, const Workspace& ws, hipStream_t s);
void launch_msm_w(size_t n, const VkTables* d_tab, const InstTab* inst_tab, const Workspace& ws, hipStream_t s);
void launch_setup_instances(const VkRaw* d_raw, const InstConsts& k, const InstRaw* d_in, InstTab* d_out, uint32_t n_inst, hipStream_t s);
void launch_vk_x(size_t n, const VkTables* d_tab, const Msm16& m16, const InstTab* inst_tab, const uint32_t* inst, const uint8_t* sig, uint8_t* out, hipStream_t s);
// lane-pair variants (k_pair.hip): one proof per two lanes, two waves per SIMD
void launch_g2chk2(size_t n, const Workspace& ws, uint8_t* status, hipStream_t s);
void launch_miller2(size_t n, const VkTables* d_tab, const Workspace& ws, uint8_t* status, hipStream_t s);
void launch_finalexp2(size_t n, const Workspace& ws, uint8_t* status, hipStream_t s);
// coefficient-parallel small-batch variants (k_wide.hip): one proof per 16 lanes
void launch_miller_w(size_t n, const VkTables* d_tab, const Workspace& ws, hipStream_t s);
void launch_finalexp_w(size_t n, const Workspace& ws, uint8_t* status, hipStream_t s);
// one proof per wavefront (k_wide.hip, four slices of 16 lanes): the smallest chunks
void launch_miller_w64(size_t n, const VkTables* d_tab, const Workspace& ws, hipStream_t s);
void launch_finalexp_w64(size_t n, const Workspace& ws, uint8_t* status, hipStream_t s);
// two wavefronts per proof: one steps the running point and tabulates the lines, the other accumulates f (the very smallest chunks)
void launch_miller_w64d(size_t n, const VkTables* d_tab, const Workspace& ws, uint8_t* status, hipStream_t s);

// mixed batches (k_mixed.hip): per-proof VM tag, device-side demultiplexing into two homogeneous sub-batches
struct MixedArgs {
    size_t n;
    const uint8_t* vm;
    const uint8_t* seals; const uint64_t* seal_off; uint32_t seal_stride;     // ragged (off) or fixed stride
    const uint8_t* in_a;                                                       // n x 32
    const uint8_t* in_b; const uint64_t* b_off; uint32_t b_stride, pv_len;     // ragged (off) or fixed stride + fixed SP1 length
    const uint32_t* cnt; const uint32_t* totals;
    uint32_t* pos; uint32_t* idx;                                              // pos[i] = slot (0xFFFFFFFF: unknown VM); idx[slot] = i
    uint8_t* c_seals; uint32_t* c_len; uint8_t* c_a; uint8_t* c_b; uint64_t* c_pvoff; uint32_t* c_pvlen;   // compact records
    uint8_t* status; uint8_t* recv;                                            // caller's outputs (unknown-VM proofs are answered here)
};
void launch_mixed_partition(const MixedArgs& a, uint32_t* cnt, uint32_t* totals, hipStream_t s);
void launch_mixed_return(size_t m, const uint32_t* idx, const uint8_t* c_status, const uint8_t* c_recv, uint8_t* status, uint8_t* recv, hipStream_t s);

// SP1 PLONK path (k_plonk.hip, zkv_plonk.h)
#ifndef ZKV_PLONK_PROOF_BYTES
#define ZKV_PLONK_PROOF_BYTES 868    /* selector + the 27 words of gnark's MarshalSolidity with one BSB22 commitment */
#endif
struct PlonkKeyRaw; struct PlonkKey;
void launch_plonk_setup(const PlonkKeyRaw* d_raw, PlonkKey* d_key, hipStream_t s);
void launch_plonk_prep(const PrepArgs& a, const PlonkKey* d_key, const Workspace& ws, hipStream_t s);

// k_wide.hip: consumer wavefronts that timed out waiting for their producer on the current device (always 0 unless a wavefront died)
int read_wait_faults(unsigned long long* out);
// multiplication-rate microbenchmark (k_diag.hip)
void launch_diag_mulmod(int kind, unsigned blocks, uint32_t iters, uint32_t* out, unsigned long long* clk, hipStream_t s);
void launch_diag_issue(int kind, unsigned blocks, uint32_t iters, uint32_t* out, unsigned long long* clk, hipStream_t s);

// aggregate check (k_agg.hip, k_pair.hip; zkv_agg.h)
struct AggTables; struct AggSeed;
void launch_setup_agg(const VkRaw* d_raw, const VkTables* d_tab, AggTables* d_agg, hipStream_t s);
void launch_agg_g1(size_t n, const VkTables* d_tab, const InstTab* inst_tab, const Workspace& ws, uint32_t* agg, const AggSeed& seed, bool sums, hipStream_t s);
void launch_agg_reduce(size_t n, uint32_t sub, bool sums, uint32_t g, const VkTables* d_tab, const Workspace& ws, const uint32_t* agg, const AggTables* tab,
                       const Workspace& ws2, uint8_t* status2, bool park, hipStream_t s);
void launch_agg_combine(size_t n64, size_t n2, uint32_t wide, const AggTables* tab, const Workspace& ws2, uint8_t* status2, hipStream_t s);
void launch_agg_miller(size_t n, uint32_t g, const VkTables* d_tab, const Workspace& ws, uint8_t* status, hipStream_t s);
void launch_agg_fprod(size_t n, size_t n2, uint32_t sub, uint32_t g, const Workspace& ws, const uint32_t* agg, const Workspace& ws2, hipStream_t s);
void launch_agg_mark(size_t n, uint32_t sub, uint32_t g, const Workspace& ws, const uint32_t* agg, const uint8_t* status2, uint8_t* status, unsigned long long* counters,
                     uint32_t* idx, hipStream_t s);
void launch_agg_gather(size_t n, const Workspace& ws, const uint32_t* agg, const unsigned long long* counters, const uint32_t* idx, const Workspace& ws3,
                       uint8_t* status3, hipStream_t s);
void launch_agg_plonk_g1(size_t n, const Workspace& ws, uint32_t* agg, const AggSeed& seed, hipStream_t s);
void launch_agg_plonk_norm(size_t n, const Workspace& ws, const uint32_t* agg, const unsigned long long* counters, const uint32_t* idx, const Workspace& ws3,
                           uint8_t* status3, hipStream_t s);
void launch_agg_scatter(size_t n, const unsigned long long* counters, const uint32_t* idx, const uint8_t* status3, uint8_t* status, hipStream_t s);

// precompile-level batches (k_precompile.hip)
void launch_ecadd(size_t n, const uint8_t* in, uint8_t* out, uint8_t* ok, hipStream_t s);
void launch_ecmul(size_t n, const uint8_t* in, uint8_t* out, uint8_t* ok, hipStream_t s);
void launch_pairing(size_t n, uint32_t k, const uint8_t* in, const Workspace& ws, uint8_t* result, uint8_t* ok, hipStream_t s);
uint32_t pairing_group(uint32_t k);      // pairs per call of the one-loop kernel (k_pairing_miller_g), 0 = none: the workspace then needs k slots per call
// the same for small batches of calls: one call per workgroup of two wavefronts (k_wide.hip)
void launch_pairing_w(size_t n, uint32_t k, const uint8_t* in, const Workspace& ws, uint8_t* result, uint8_t* ok, hipStream_t s);

}  // namespace zkv
