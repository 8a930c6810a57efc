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
    for (int k = 0; k < 8; k++) {
        ws.prep[(size_t)(64 + k) * ws.cap + i] = o.s[0][k];
        ws.prep[(size_t)(72 + k) * ws.cap + i] = o.s[1][k];
    }
}

// shared front: locate the record, run the reference's ordered checks (len < 4, selector, strict decode).
// Returns true when the 8 words should be parsed; otherwise *st holds the final status.
__device__ __forceinline__ bool front_checks(const PrepArgs& a, size_t i, const uint8_t*& rec, uint8_t& st) {
    size_t len;
    if (a.off) { rec = a.blob + a.off[i]; len = (size_t)(a.off[i + 1] - a.off[i]); }
    else { rec = a.blob + i * (size_t)a.stride; len = a.stride; }
    uint32_t rv = 0;
    bool go = false;
    if (len < 4) st = ST_INVALID_PROOF_DATA;                       // verifier.rs:151 / sp1 verifier.rs:64
    else {
        uint32_t sel = load_be32(rec);
        if (sel != a.selector_be) { st = ST_SELECTOR_MISMATCH; rv = sel; }        // :155-165 / :68-78
        else if (len != 260) st = ST_INVALID_PROOF_DATA;            // strict abi_decode of 8 static words
        else if (a.force_fail) st = ST_VERIFICATION_FAILED;
        else { st = ST_VERIFICATION_FAILED; go = true; }
    }
    if (a.recv) {
        a.recv[4 * i] = (uint8_t)(rv >> 24); a.recv[4 * i + 1] = (uint8_t)(rv >> 16);
        a.recv[4 * i + 2] = (uint8_t)(rv >> 8); a.recv[4 * i + 3] = (uint8_t)rv;
    }
    return go;
}

__global__ __launch_bounds__(ZKV_BLOCK) void k_prep_risc0(PrepArgs a, Risc0Consts k, Workspace ws) {
    size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const uint8_t* rec; uint8_t st;
    uint32_t flags = 0;
    if (front_checks(a, i, rec, st)) {
        PrepOut o;
        uint32_t h[8];
        if (a.in32_b) risc0_claim_digest(k, a.in32_a + 32 * i, a.in32_b + 32 * i, h);
        else {
#pragma unroll 1
            for (int j = 0; j < 8; j++) h[j] = load_be32(a.in32_a + 32 * i + 4 * j);
        }
        risc0_split_digest(h, o.s[0], o.s[1]);                      // 128-bit halves: always < R
        uint32_t w[8][8];
#pragma unroll 1
        for (int j = 0; j < 8; j++) load_be256(w[j], rec + 4 + 32 * j);
        if (prep_points(w, true, o)) { flags = o.flags; store_prep(ws, i, o); }
    }
    ws.flags[i] = flags;
    a.status[i] = st;
}

__global__ __launch_bounds__(ZKV_BLOCK) void k_prep_sp1(PrepArgs a, Workspace ws) {
    size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const uint8_t* rec; uint8_t st;
    uint32_t flags = 0;
    if (front_checks(a, i, rec, st)) {
        PrepOut o;
        load_be256(o.s[0], a.in32_a + 32 * i);                      // U256::from_be_bytes(program_vkey), sp1/types.rs:24
        const uint8_t* pv; size_t pvl;
        if (a.pv_off) { pv = a.pv_blob + a.pv_off[i]; pvl = (size_t)(a.pv_off[i + 1] - a.pv_off[i]); }
        else { pv = a.pv_blob + i * (size_t)a.pv_stride; pvl = a.pv_stride; }
        uint32_t h[8];
        sha256_bytes(pv, pvl, h);
        h[0] &= 0x1fffffffu;                                        // & (2^253 - 1); % R is the identity below R
#pragma unroll
        for (int j = 0; j < 8; j++) o.s[1][7 - j] = h[j];
        if (raw_lt_r(o.s[0])) {                                     // groth16.rs:32
            uint32_t w[8][8];
#pragma unroll 1
            for (int j = 0; j < 8; j++) load_be256(w[j], rec + 4 + 32 * j);
            if (prep_points(w, false, o)) { flags = o.flags; store_prep(ws, i, o); }
        }
    }
    ws.flags[i] = flags;
    a.status[i] = st;
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
