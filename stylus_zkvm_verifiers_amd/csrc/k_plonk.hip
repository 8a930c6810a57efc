// SP1 PLONK path: stage PREP of a PLONK batch.  One proof per lane: the reference's ordered checks for
// `ISp1Verifier::verify_proof` (sp1/verifier.rs:58-111: length < 4, selector, strict length), the two public inputs
// (sp1/types.rs:22-38), then everything of the gnark-style verification that precedes the pairing (zkv_plonk.h).  The two G1
// points of the final 2-pair check are written where the Groth16 pipeline keeps L = vk_x and C, so the lane-pair Miller loop and
// final exponentiation run unchanged with the SRS's G2 points as the fixed pairs.  Parity unpinned by construction (no PLONK in
// the reference).
#include "zkv_internal.h"
#include "zkv_plonk.h"

namespace zkv {

__global__ __launch_bounds__(64) void k_plonk_setup(const PlonkKeyRaw* __restrict__ raw, PlonkKey* __restrict__ key) {
    if (blockIdx.x == 0 && threadIdx.x == 0) plonk_setup_key(*raw, *key);
}

#ifndef ZKV_PLONK_WAVES
#define ZKV_PLONK_WAVES 4        /* measured on 2^18 proofs: 198 ms at one wave per SIMD, 150 at two, 143 at three, 141 at four */
#endif
__global__ __launch_bounds__(ZKV_BLOCK, ZKV_PLONK_WAVES) void k_plonk_prep(PrepArgs a, const PlonkKey* __restrict__ key, Workspace ws) {
    size_t i = (size_t)blockIdx.x * ZKV_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const uint8_t* rec; size_t len;
    if (a.off) { rec = a.blob + a.off[i]; len = (size_t)(a.off[i + 1] - a.off[i]); }
    else { rec = a.blob + i * (size_t)a.stride; len = a.stride; }
    uint8_t st = ST_VERIFICATION_FAILED;
    uint32_t rv = 0, flags = 0;
    bool go = false;
    if (len < 4) st = ST_INVALID_PROOF_DATA;                                        // sp1/verifier.rs:64
    else {
        const uint32_t sel = load_be32(rec);
        if (sel != a.selector_be) { st = ST_SELECTOR_MISMATCH; rv = sel; }           // :68-78
        else if (len != ZKV_PLONK_PROOF_BYTES) st = ST_INVALID_PROOF_DATA;           // strict decode of the 27 static words
        else go = !a.force_fail;
    }
    if (a.recv) {
        a.recv[4 * i] = (uint8_t)(rv >> 24); a.recv[4 * i + 1] = (uint8_t)(rv >> 16);
        a.recv[4 * i + 2] = (uint8_t)(rv >> 8); a.recv[4 * i + 3] = (uint8_t)rv;
    }
    if (go) {
        uint32_t w[27][8], pub[2][8];
#pragma unroll 1
        for (int k = 0; k < 27; k++) load_be256(w[k], rec + 4 + 32 * k);
        load_be256(pub[0], a.in32_a + 32 * i);                                      // U256::from_be_bytes(program_vkey), sp1/types.rs:24
        const uint8_t* pv; size_t pvl;
        if (a.pv_off) { pv = a.pv_blob + a.pv_off[i]; pvl = (size_t)(a.pv_off[i + 1] - a.pv_off[i]); }
        else { pv = a.pv_blob + i * (size_t)a.pv_stride; pvl = a.pv_stride; }
        uint32_t h[8];
        sha256_bytes(pv, pvl, h);
        h[0] &= 0x1fffffffu;                                                        // & (2^253 - 1), sp1/types.rs:34-38
        for (int j = 0; j < 8; j++) pub[1][7 - j] = h[j];
        PlonkOut o;
        const TabRef tab = {a.plonk_tab + i * (size_t)PLONK_TAB_WORDS};             // this proof's contiguous table region (3,840 bytes)
        if (plonk_prepare(*key, w, pub, o, tab)) {
            // x/y = X Z / Y and 1/y = Z^3 / Y of the two points, as the Miller loop's fixed pairs expect them: one inversion for both
            const Fp one = fp_one();
            const bool d_inf = fp_is_zero(o.d.z), q_inf = fp_is_zero(o.q.z);
            const Fp yd = d_inf ? one : o.d.y, yq = q_inf ? one : o.q.y;
            const Fp inv = fp_inv(fp_mul(yd, yq));
            const Fp iyd = fp_mul(inv, yq), iyq = fp_mul(inv, yd);
            const Fp z = fp_zero();
            ws_st(ws.norm, ws.cap, 0, i, z); ws_st(ws.norm, ws.cap, 8, i, z);
            ws_st(ws.norm, ws.cap, 16, i, fp_mul(fp_mul(o.d.x, o.d.z), iyd)); ws_st(ws.norm, ws.cap, 24, i, fp_mul(fp_mul(fp_sqr(o.d.z), o.d.z), iyd));
            ws_st(ws.norm, ws.cap, 32, i, fp_mul(fp_mul(o.q.x, o.q.z), iyq)); ws_st(ws.norm, ws.cap, 40, i, fp_mul(fp_mul(fp_sqr(o.q.z), o.q.z), iyq));
            flags = FL_ALIVE | FL_A_INF | FL_B_INF | (d_inf ? FL_L_INF : 0u) | (q_inf ? FL_C_INF : 0u);
        }
    }
    ws.flags[i] = flags;
    ws.g2bad[i] = 0;
    a.status[i] = st;
}

// the multiples of one point per block (one lane works: the levels of the row are sequential), then one lane per row (point, a) of the
// joint P / phi(P) tables
__global__ __launch_bounds__(64) void k_plonk_mult(PlonkKey* __restrict__ key) {
    if (threadIdx.x == 0 && blockIdx.x <= PK_POINTS) plonk_setup_mult(*key, (int)blockIdx.x);
}
__global__ __launch_bounds__(64) void k_plonk_joint(PlonkKey* __restrict__ key) {
    const int t = (int)(blockIdx.x * 64 + threadIdx.x);
    if (t < (PK_POINTS + 1) * (PK_JA + 1)) plonk_joint_row(*key, t / (PK_JA + 1), t % (PK_JA + 1));
}
void launch_plonk_setup(const PlonkKeyRaw* d_raw, PlonkKey* d_key, hipStream_t s) {
    hipLaunchKernelGGL(k_plonk_setup, dim3(1), dim3(64), 0, s, d_raw, d_key);
    hipLaunchKernelGGL(k_plonk_mult, dim3(PK_POINTS + 1), dim3(64), 0, s, d_key);
    hipLaunchKernelGGL(k_plonk_joint, dim3(((PK_POINTS + 1) * (PK_JA + 1) + 63) / 64), dim3(64), 0, s, d_key);
}
void launch_plonk_prep(const PrepArgs& a, const PlonkKey* d_key, const Workspace& ws, hipStream_t s) {
    if (!a.n) return;
    hipLaunchKernelGGL(k_plonk_prep, dim3((unsigned)((a.n + ZKV_BLOCK - 1) / ZKV_BLOCK)), dim3(ZKV_BLOCK), 0, s, a, d_key, ws);
}

}  // namespace zkv
