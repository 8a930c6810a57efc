// Stage PREP: seal parsing, selector / strict-length checks, SHA-256 digest chain, signal range checks,
// A-negation, coordinate / on-curve validation and conversion to Montgomery form.  One proof per lane.
#include "zkv_internal.h"

namespace zkv {

__device__ __forceinline__ void store_prep(const Workspace& ws, size_t i, const PrepOut& o) {
    ws_st(ws.prep, ws.cap, 0, i, o.ax); ws_st(ws.prep, ws.cap, 8, i, o.ay);
    ws_st(ws.prep, ws.cap, 16, i, o.cx); ws_st(ws.prep, ws.cap, 24, i, o.cy);
    ws_st(ws.prep, ws.cap, 32, i, o.bx.c0); ws_st(ws.prep, ws.cap, 40, i, o.bx.c1);
    ws_st(ws.prep, ws.cap, 48, i, o.by.c0); ws_st(ws.prep, ws.cap, 56, i, o.by.c1);
#pragma unroll
    for (int b = 0; b < MAX_VAR; b++) {
#pragma unroll
        for (int k = 0; k < 8; k++) ws.prep[(size_t)(64 + 8 * b + k) * ws.cap + i] = o.s[b][k];
    }
}

// Fixed-stride batches (the HBM-resident fast path): the 64 seals of a wavefront are 16,640 contiguous bytes; the wave
// copies them to LDS with fully coalesced dword loads (256 B per wave-instruction) and every lane then reads its own
// 65 words from LDS (row stride 65 dwords: conflict-free).  Ragged batches read their records straight from HBM.
struct SealReader {
    const uint32_t* lds_row;     // non-null: staged record of this lane (words are still big-endian bytes)
    const uint8_t* rec;          // otherwise: the record in HBM
    __device__ __forceinline__ uint32_t word(int k) const {
        return lds_row ? __builtin_bswap32(lds_row[k]) : load_be32(rec + 4 * k);
    }
    __device__ __forceinline__ void u256(uint32_t limbs[8], int word0) const {
#pragma unroll 1
        for (int j = 0; j < 8; j++) limbs[7 - j] = word(word0 + j);
    }
};
__device__ __forceinline__ bool stage_seals(const PrepArgs& a, uint32_t* lds) {
    if (a.off || a.stride != 260 || ((uintptr_t)a.blob & 3u)) return false;          // wave-uniform
    size_t base = (size_t)blockIdx.x * ZKV_BLOCK;
    size_t m = a.n - base < ZKV_BLOCK ? a.n - base : ZKV_BLOCK;
    const uint32_t* src = (const uint32_t*)(a.blob + base * 260);
    uint32_t total = (uint32_t)m * 65u;
#pragma unroll 1
    for (uint32_t j = threadIdx.x; j < total; j += ZKV_BLOCK) lds[j] = src[j];
    __syncthreads();
    return true;
}

// shared front: locate the record, run the reference's ordered checks (len < 4, selector, strict decode).
// Returns true when the 8 words should be parsed; otherwise *st holds the final status.
__device__ __forceinline__ bool front_checks(const PrepArgs& a, size_t i, bool staged, const uint32_t* lds, SealReader& rd, uint8_t& st,
                                             uint32_t& inst_bits) {
    size_t len;
    rd.lds_row = nullptr;
    if (a.off) { rd.rec = a.blob + a.off[i]; len = (size_t)(a.off[i + 1] - a.off[i]); }
    else {
        rd.rec = a.blob + i * (size_t)a.stride; len = a.len ? a.len[i] : a.stride;
        if (staged) rd.lds_row = lds + threadIdx.x * 65u;
    }
    uint32_t rv = 0, expect = a.selector_be, ctx_fail = a.force_fail;
    bool go = false, known = true;
    inst_bits = 0;
    if (a.inst) {                                                   // verifier set: this proof's instance
        const uint32_t idx = a.inst[i];
        known = idx < a.n_inst;                                     // an unregistered instance is an un-initialised verifier
        if (known) { expect = a.inst_tab[idx].selector_be; ctx_fail = a.inst_tab[idx].fail; inst_bits = idx << 8; }
    }
    if (a.len && len == 0xFFFFFFFFu) st = ST_BAD_CALLDATA;         // wire layer: the router could not decode the call
    else if (a.not_initialized || !known) st = ST_INVALID_INITIALIZATION;     // risc0/verifier.rs:84-86, 99-101
    else if (len < 4) st = ST_INVALID_PROOF_DATA;                       // verifier.rs:151 / sp1 verifier.rs:64
    else {
        uint32_t sel = rd.word(0);
        if (sel != expect) { st = ST_SELECTOR_MISMATCH; rv = sel; }        // :155-165 / :68-78
        else if (len != 260) st = ST_INVALID_PROOF_DATA;            // strict abi_decode of 8 static words
        else if (ctx_fail) st = ST_VERIFICATION_FAILED;
        else { st = ST_VERIFICATION_FAILED; go = true; }
    }
    if (a.recv) {
        a.recv[4 * i] = (uint8_t)(rv >> 24); a.recv[4 * i + 1] = (uint8_t)(rv >> 16);
        a.recv[4 * i + 2] = (uint8_t)(rv >> 8); a.recv[4 * i + 3] = (uint8_t)rv;
    }
    return go;
}

__global__ __launch_bounds__(ZKV_BLOCK) void k_prep_risc0(PrepArgs a, Risc0Consts k, Workspace ws) {
    __shared__ uint32_t seal_lds[ZKV_BLOCK * 65];
    const bool staged = stage_seals(a, seal_lds);
    size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    SealReader rd; uint8_t st;
    uint32_t flags = 0, inst_bits;
    if (front_checks(a, i, staged, seal_lds, rd, st, inst_bits)) {
        PrepOut o;
        for (int b = 2; b < MAX_VAR; b++) for (int k2 = 0; k2 < 8; k2++) o.s[b][k2] = 0;
        uint32_t h[8];
        const bool integrity = a.kind ? a.kind[i] != 0 : a.in32_b == nullptr;
        if (!integrity) risc0_claim_digest(k, a.in32_a + 32 * i, a.in32_b + 32 * i, h);
        else {
#pragma unroll 1
            for (int j = 0; j < 8; j++) h[j] = load_be32(a.in32_a + 32 * i + 4 * j);
        }
        risc0_split_digest(h, o.s[0], o.s[1]);                      // 128-bit halves: always < R
        uint32_t w[8][8];
#pragma unroll 1
        for (int j = 0; j < 8; j++) rd.u256(w[j], 1 + 8 * j);
        if (prep_points(w, true, o)) { flags = o.flags | inst_bits; store_prep(ws, i, o); }
    }
    ws.flags[i] = flags;
    ws.g2bad[i] = 0;
    a.status[i] = st;
}

__global__ __launch_bounds__(ZKV_BLOCK) void k_prep_sp1(PrepArgs a, Workspace ws) {
    __shared__ uint32_t seal_lds[ZKV_BLOCK * 65];
    const bool staged = stage_seals(a, seal_lds);
    size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    SealReader rd; uint8_t st;
    uint32_t flags = 0, inst_bits;
    if (front_checks(a, i, staged, seal_lds, rd, st, inst_bits)) {
        PrepOut o;
        for (int b = 2; b < MAX_VAR; b++) for (int k2 = 0; k2 < 8; k2++) o.s[b][k2] = 0;
        load_be256(o.s[0], a.in32_a + 32 * i);                      // U256::from_be_bytes(program_vkey), sp1/types.rs:24
        const uint8_t* pv; size_t pvl;
        if (a.pv_len) { pv = a.pv_blob + a.pv_off[i]; pvl = a.pv_len[i]; }
        else if (a.pv_off) { pv = a.pv_blob + a.pv_off[i]; pvl = (size_t)(a.pv_off[i + 1] - a.pv_off[i]); }
        else { pv = a.pv_blob + i * (size_t)a.pv_stride; pvl = a.pv_stride; }
        uint32_t h[8];
        sha256_bytes(pv, pvl, h);
        h[0] &= 0x1fffffffu;                                        // & (2^253 - 1); % R is the identity below R
#pragma unroll
        for (int j = 0; j < 8; j++) o.s[1][7 - j] = h[j];
        if (raw_lt_r(o.s[0])) {                                     // groth16.rs:32
            uint32_t w[8][8];
#pragma unroll 1
            for (int j = 0; j < 8; j++) rd.u256(w[j], 1 + 8 * j);
            if (prep_points(w, false, o)) { flags = o.flags; store_prep(ws, i, o); }
        }
    }
    ws.flags[i] = flags;
    ws.g2bad[i] = 0;
    a.status[i] = st;
}

// Groth16Verifier::verify_proof_with_key for an arbitrary key (common/groth16.rs:23-49): 8 proof words + n_sig signals per
// proof, no selector / hashing.  status 0 <=> the function returns true.
__global__ __launch_bounds__(ZKV_BLOCK) void k_prep_groth16(PrepArgs a, Workspace ws) {
    size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    uint32_t flags = 0;
    if (!a.force_fail) {
        PrepOut o;
        bool ok = true;
#pragma unroll 1
        for (uint32_t b = 0; b < MAX_VAR; b++) {
            if (b < a.n_sig) { load_be256(o.s[b], a.in32_a + 32 * ((size_t)a.n_sig * i + b)); ok = ok && raw_lt_r(o.s[b]); }   // groth16.rs:32
            else for (int k = 0; k < 8; k++) o.s[b][k] = 0;
        }
        if (ok) {
            uint32_t w[8][8];
            const uint8_t* rec = a.blob + 256 * i;
#pragma unroll 1
            for (int j = 0; j < 8; j++) load_be256(w[j], rec + 32 * j);
            if (prep_points(w, a.negate_a != 0, o)) { flags = o.flags; store_prep(ws, i, o); }
        }
    }
    ws.flags[i] = flags;
    ws.g2bad[i] = 0;
    a.status[i] = ST_VERIFICATION_FAILED;
}

void launch_prep_groth16(const PrepArgs& a, const Workspace& ws, hipStream_t s) {
    if (!a.n) return;
    hipLaunchKernelGGL(k_prep_groth16, dim3((unsigned)((a.n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, a, ws);
}
void launch_prep_risc0(const PrepArgs& a, const Risc0Consts& k, const Workspace& ws, hipStream_t s) {
    if (!a.n) return;
    hipLaunchKernelGGL(k_prep_risc0, dim3((unsigned)((a.n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, a, k, ws);
}
void launch_prep_sp1(const PrepArgs& a, const Workspace& ws, hipStream_t s) {
    if (!a.n) return;
    hipLaunchKernelGGL(k_prep_sp1, dim3((unsigned)((a.n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, a, ws);
}

}  // namespace zkv
