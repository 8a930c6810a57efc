// SHA-256 for the digest chains of the verify path (the reference uses the `sha2` crate in WASM:
// /root/reference/contracts/src/risc0/types.rs:62-94, sp1/types.rs:34-38).  One message per lane.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "zkv_field.h"

namespace zkv {

ZKV_HD uint32_t ror32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

ZKV_HD void sha256_init(uint32_t h[8]) {
    h[0] = 0x6a09e667u; h[1] = 0xbb67ae85u; h[2] = 0x3c6ef372u; h[3] = 0xa54ff53au;
    h[4] = 0x510e527fu; h[5] = 0x9b05688cu; h[6] = 0x1f83d9abu; h[7] = 0x5be0cd19u;
}

// one compression; w[16] is the big-endian-decoded block and is clobbered
ZKV_HD void sha256_compress(uint32_t h[8], uint32_t w[16]) {
    const uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
        0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
        0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
        0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
        0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
        0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
        0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
        0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            uint32_t s0 = ror32(w15, 7) ^ ror32(w15, 18) ^ (w15 >> 3);
            uint32_t s1 = ror32(w2, 17) ^ ror32(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
        }
        uint32_t t1 = hh + (ror32(e, 6) ^ ror32(e, 11) ^ ror32(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i & 15];
        uint32_t t2 = (ror32(a, 2) ^ ror32(a, 13) ^ ror32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

ZKV_HD uint32_t load_be32(const uint8_t* p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

// generic message of `len` bytes at `msg` -> digest words (big-endian word order, h[0] first)
ZKV_HD void sha256_bytes(const uint8_t* msg, size_t len, uint32_t h[8]) {
    uint32_t w[16];
    sha256_init(h);
    size_t off = 0;
#pragma unroll 1
    for (; off + 64 <= len; off += 64) {
#pragma unroll 1
        for (int i = 0; i < 16; i++) w[i] = load_be32(msg + off + 4 * i);
        sha256_compress(h, w);
    }
    size_t rem = len - off;
    // tail: remaining bytes, 0x80, zero pad, 64-bit bit length
#pragma unroll 1
    for (int i = 0; i < 16; i++) {
        uint32_t v = 0;
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
            size_t pos = (size_t)(4 * i + k);
            uint32_t byte = pos < rem ? msg[off + pos] : (pos == rem ? 0x80u : 0u);
            v = (v << 8) | byte;
        }
        w[i] = v;
    }
    uint64_t bits = (uint64_t)len * 8u;
    if (rem + 9 <= 64) {
        w[14] = (uint32_t)(bits >> 32); w[15] = (uint32_t)bits;
        sha256_compress(h, w);
    } else {
        sha256_compress(h, w);
#pragma unroll 1
        for (int i = 0; i < 14; i++) w[i] = 0;
        w[14] = (uint32_t)(bits >> 32); w[15] = (uint32_t)bits;
        sha256_compress(h, w);
    }
}

}  // namespace zkv
