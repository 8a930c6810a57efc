// Mixed batches: every proof names its VM (`VMType`, /root/reference/contracts/src/common/types.rs:24-26; the reference
// passes it to Groth16Verifier::verify_proof_with_key per call, common/groth16.rs:23-31, 96-103).
//
// The two VMs differ in everything but the arithmetic (digest chain, selector, key tables, A negation), and the line /
// window tables are wave-uniform reads in the Miller and MSM kernels, so a mixed batch is first DEMULTIPLEXED on the
// device into two homogeneous sub-batches, each of which then takes the ordinary stage pipeline of its own verifier:
//   k_mixed_count   per 256-proof block: how many RISC Zero / SP1 tags
//   k_mixed_scan    exclusive scan of the block counts (one workgroup), totals n0 / n1
//   k_mixed_place   stable partition: pos[i] = slot of proof i in the compact order (RISC Zero first, then SP1)
//   k_mixed_gather  copies seal, 32-byte input(s) and the public-values location of proof i to slot pos[i]
//   k_mixed_return  status / received selector of slot j go back to proof idx[j]
// All of it is byte traffic (about 2 x 400 B per proof), negligible beside the pairing.
#include "zkv_internal.h"

namespace zkv {

constexpr int MX_BLOCK = 256;

__device__ __forceinline__ int mx_class(uint8_t tag) { return tag == 0 ? 0 : tag == 1 ? 1 : 2; }   // ZKV_VM_RISC0, ZKV_VM_SP1, unknown

__global__ __launch_bounds__(MX_BLOCK) void k_mixed_count(size_t n, const uint8_t* __restrict__ vm, uint32_t* __restrict__ cnt) {
    size_t i = (size_t)blockIdx.x * MX_BLOCK + threadIdx.x;
    const int c = i < n ? mx_class(vm[i]) : 2;
    const int c0 = __syncthreads_count(c == 0), c1 = __syncthreads_count(c == 1);
    if (threadIdx.x == 0) { cnt[2 * blockIdx.x] = (uint32_t)c0; cnt[2 * blockIdx.x + 1] = (uint32_t)c1; }
}

// cnt[2b], cnt[2b+1] -> exclusive prefix sums in place; totals[0] = n0, totals[1] = n1.  One workgroup.
__global__ __launch_bounds__(1024) void k_mixed_scan(uint32_t blocks, uint32_t* __restrict__ cnt, uint32_t* __restrict__ totals) {
    __shared__ uint32_t part[2][1024];
    const uint32_t per = (blocks + 1023u) / 1024u, lo = threadIdx.x * per, hi = lo + per < blocks ? lo + per : blocks;
    uint32_t s0 = 0, s1 = 0;
    for (uint32_t b = lo; b < hi; b++) { s0 += cnt[2 * b]; s1 += cnt[2 * b + 1]; }
    part[0][threadIdx.x] = s0; part[1][threadIdx.x] = s1;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {                       // Hillis-Steele inclusive scan of the 1024 partial sums
        uint32_t a0 = threadIdx.x >= d ? part[0][threadIdx.x - d] : 0, a1 = threadIdx.x >= d ? part[1][threadIdx.x - d] : 0;
        __syncthreads();
        part[0][threadIdx.x] += a0; part[1][threadIdx.x] += a1;
        __syncthreads();
    }
    uint32_t e0 = part[0][threadIdx.x] - s0, e1 = part[1][threadIdx.x] - s1;
    for (uint32_t b = lo; b < hi; b++) {
        uint32_t c0 = cnt[2 * b], c1 = cnt[2 * b + 1];
        cnt[2 * b] = e0; cnt[2 * b + 1] = e1;
        e0 += c0; e1 += c1;
    }
    if (threadIdx.x == 1023) { totals[0] = part[0][1023]; totals[1] = part[1][1023]; }
}


__global__ __launch_bounds__(MX_BLOCK) void k_mixed_place(MixedArgs a) {
    __shared__ uint32_t wave_cnt[2][MX_BLOCK / 64];
    size_t i = (size_t)blockIdx.x * MX_BLOCK + threadIdx.x;
    const int c = i < a.n ? mx_class(a.vm[i]) : 2;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t m0 = __ballot(c == 0), m1 = __ballot(c == 1);
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    if (lane == 0) { wave_cnt[0][wave] = (uint32_t)__popcll(m0); wave_cnt[1][wave] = (uint32_t)__popcll(m1); }
    __syncthreads();
    if (i >= a.n) return;
    uint32_t r = (uint32_t)__popcll((c == 0 ? m0 : m1) & below);
    for (uint32_t w = 0; w < wave; w++) r += wave_cnt[c == 0 ? 0 : 1][w];
    uint32_t slot = 0xFFFFFFFFu;
    if (c == 0) slot = a.cnt[2 * blockIdx.x] + r;
    else if (c == 1) slot = a.totals[0] + a.cnt[2 * blockIdx.x + 1] + r;
    a.pos[i] = slot;
    if (c == 2) {                                           // not a VMType: no verifier to ask
        a.status[i] = 7;                                    // ZKV_STATUS_UNKNOWN_VM
        if (a.recv) { a.recv[4 * i] = 0; a.recv[4 * i + 1] = 0; a.recv[4 * i + 2] = 0; a.recv[4 * i + 3] = 0; }
        return;
    }
    a.idx[slot] = (uint32_t)i;
    // per-proof scalars of the compact record
    size_t len = a.seal_off ? (size_t)(a.seal_off[i + 1] - a.seal_off[i]) : a.seal_stride;
    a.c_len[slot] = len > 0xFFFFFFFEu ? 0xFFFFFFFEu : (uint32_t)len;
    if (c == 1) {
        const uint64_t start = a.b_off ? a.b_off[i] : (uint64_t)i * a.b_stride;
        a.c_pvoff[slot] = start;
        a.c_pvlen[slot] = a.b_off ? (uint32_t)(a.b_off[i + 1] - a.b_off[i]) : a.pv_len;
    } else { a.c_pvoff[slot] = 0; a.c_pvlen[slot] = 0; }
}

// One thread per (proof, word): 65 seal words, 8 words of in_a, 8 words of in_b (the RISC Zero journal digest).
constexpr uint32_t MX_WORDS = 65 + 8 + 8;
__device__ __forceinline__ uint32_t mx_ld4(const uint8_t* p, size_t avail) {      // up to 4 bytes, zero padded, any alignment
    if (avail >= 4 && !((uintptr_t)p & 3u)) return *(const uint32_t*)p;
    uint32_t v = 0;
    for (int k = 0; k < 4; k++) if ((size_t)k < avail) v |= (uint32_t)p[k] << (8 * k);
    return v;
}
__global__ __launch_bounds__(MX_BLOCK) void k_mixed_gather(MixedArgs a) {
    const size_t t = (size_t)blockIdx.x * MX_BLOCK + threadIdx.x;
    const size_t i = t / MX_WORDS;
    const uint32_t w = (uint32_t)(t - i * MX_WORDS);
    if (i >= a.n) return;
    const uint32_t slot = a.pos[i];
    if (slot == 0xFFFFFFFFu) return;
    if (w < 65) {
        const uint8_t* src = a.seal_off ? a.seals + a.seal_off[i] : a.seals + i * (size_t)a.seal_stride;
        size_t len = a.seal_off ? (size_t)(a.seal_off[i + 1] - a.seal_off[i]) : a.seal_stride;
        if (len > 260) len = 260;
        const size_t at = 4u * w;
        ((uint32_t*)a.c_seals)[(size_t)slot * 65 + w] = at < len ? mx_ld4(src + at, len - at) : 0u;
    } else if (w < 73) {
        ((uint32_t*)a.c_a)[(size_t)slot * 8 + (w - 65)] = mx_ld4(a.in_a + 32 * i + 4 * (w - 65), 4);
    } else if (a.vm[i] == 0) {
        const uint8_t* src = a.b_off ? a.in_b + a.b_off[i] : a.in_b + i * (size_t)a.b_stride;
        ((uint32_t*)a.c_b)[(size_t)slot * 8 + (w - 73)] = mx_ld4(src + 4 * (w - 73), 4);
    }
}

__global__ __launch_bounds__(MX_BLOCK) void k_mixed_return(size_t m, const uint32_t* __restrict__ idx, const uint8_t* __restrict__ c_status,
                                                            const uint8_t* __restrict__ c_recv, uint8_t* __restrict__ status, uint8_t* __restrict__ recv) {
    size_t j = (size_t)blockIdx.x * MX_BLOCK + threadIdx.x;
    if (j >= m) return;
    const uint32_t i = idx[j];
    status[i] = c_status[j];
    if (recv) {
#pragma unroll
        for (int k = 0; k < 4; k++) recv[4 * (size_t)i + k] = c_recv[4 * j + k];
    }
}

void launch_mixed_partition(const MixedArgs& a, uint32_t* cnt, uint32_t* totals, hipStream_t s) {
    if (!a.n) return;
    const unsigned blocks = (unsigned)((a.n + MX_BLOCK - 1) / MX_BLOCK);
    hipLaunchKernelGGL(k_mixed_count, dim3(blocks), dim3(MX_BLOCK), 0, s, a.n, a.vm, cnt);
    hipLaunchKernelGGL(k_mixed_scan, dim3(1), dim3(1024), 0, s, blocks, cnt, totals);
    hipLaunchKernelGGL(k_mixed_place, dim3(blocks), dim3(MX_BLOCK), 0, s, a);
    const size_t threads = a.n * MX_WORDS;
    hipLaunchKernelGGL(k_mixed_gather, dim3((unsigned)((threads + MX_BLOCK - 1) / MX_BLOCK)), dim3(MX_BLOCK), 0, s, a);
}
void launch_mixed_return(size_t m, const uint32_t* idx, const uint8_t* c_status, const uint8_t* c_recv, uint8_t* status, uint8_t* recv, hipStream_t s) {
    if (!m) return;
    hipLaunchKernelGGL(k_mixed_return, dim3((unsigned)((m + MX_BLOCK - 1) / MX_BLOCK)), dim3(MX_BLOCK), 0, s, m, idx, c_status, c_recv, status, recv);
}

}  // namespace zkv
