// Host side of the on-chain wire layer: Solidity function selectors, calldata encoders for clients and the
// return / revert data of one eth_call.  Bulk decoding of verify calls happens on the device (k_wire.hip); this header only
// covers the constant-size pieces.
//
// Reference: the methods the example shells export -- examples/risc0-verifier/src/lib.rs (IRiscZeroVerifier,
// contracts/src/risc0/verifier.rs:18-42), examples/sp1-verifier/src/lib.rs (ISp1Verifier, contracts/src/sp1/verifier.rs:16-29)
// -- under the Solidity signatures the clients use (examples/risc0-verifier/examples/interact.rs:31-43,
// examples/sp1-verifier/examples/interact.rs:11-19; `Vec<u8>` = `uint8[]`).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace zkv {
namespace host {

// ---- Keccak-256 (Ethereum's pre-NIST padding), sponge rate 136 bytes
struct Keccak {
    uint64_t st[25];
    static uint64_t rotl(uint64_t v, unsigned n) { return n ? (v << n) | (v >> (64 - n)) : v; }
    void permute() {
        static const unsigned rho[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
        static const unsigned pi[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
        uint64_t lfsr = 1;
        for (int round = 0; round < 24; round++) {
            uint64_t par[5];
            for (int x = 0; x < 5; x++) par[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
            for (int x = 0; x < 5; x++) {
                uint64_t d = par[(x + 4) % 5] ^ rotl(par[(x + 1) % 5], 1);
                for (int y = 0; y < 25; y += 5) st[y + x] ^= d;
            }
            uint64_t carry = st[1];                       // rho + pi along the single 24-cycle of the lane permutation
            for (int t = 0; t < 24; t++) { uint64_t nx = st[pi[t]]; st[pi[t]] = rotl(carry, rho[t]); carry = nx; }
            for (int y = 0; y < 25; y += 5) {
                uint64_t row[5];
                for (int x = 0; x < 5; x++) row[x] = st[y + x];
                for (int x = 0; x < 5; x++) st[y + x] = row[x] ^ (~row[(x + 1) % 5] & row[(x + 2) % 5]);
            }
            uint64_t rc = 0;                              // round constant from the degree-8 LFSR of the specification
            for (int j = 0; j < 7; j++) {
                if (lfsr & 1) rc ^= 1ull << ((1u << j) - 1);
                lfsr = (lfsr & 0x80) ? ((lfsr << 1) ^ 0x171) : (lfsr << 1);
            }
            st[0] ^= rc;
        }
    }
};
inline void keccak256(const uint8_t* msg, size_t len, uint8_t out[32]) {
    Keccak k;
    memset(k.st, 0, sizeof k.st);
    size_t pos = 0;
    auto absorb = [&](uint8_t b) {
        k.st[pos >> 3] ^= (uint64_t)b << (8 * (pos & 7));
        if (++pos == 136) { k.permute(); pos = 0; }
    };
    for (size_t i = 0; i < len; i++) absorb(msg[i]);
    k.st[pos >> 3] ^= (uint64_t)0x01 << (8 * (pos & 7));
    k.st[16] ^= 0x8000000000000000ull;
    k.permute();
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)(k.st[i >> 3] >> (8 * (i & 7)));
}
inline void fn_selector(const char* signature, uint8_t out[4]) {
    uint8_t h[32];
    keccak256((const uint8_t*)signature, strlen(signature), h);
    memcpy(out, h, 4);
}

// ---- methods
enum Risc0Fn { R0_INITIALIZE, R0_VERIFY, R0_VERIFY_INTEGRITY, R0_IS_INITIALIZED, R0_GET_SELECTOR, R0_GET_CONTROL_ROOT, R0_GET_BN254_CONTROL_ID,
               R0_GET_VERIFIER_KEY_DIGEST, R0_COUNT };
static const char* const RISC0_SIGNATURES[R0_COUNT] = {
    "initialize(bytes32,bytes32)", "verify(uint8[],bytes32,bytes32)", "verifyIntegrity(uint8[],bytes32)", "isInitialized()", "getSelector()",
    "getControlRoot()", "getBn254ControlId()", "getVerifierKeyDigest()"};
enum Sp1Fn { SP1_VERIFY_PROOF, SP1_FN_VERIFIER_HASH, SP1_FN_VERSION, SP1_COUNT };
static const char* const SP1_SIGNATURES[SP1_COUNT] = {"verifyProof(bytes32,uint8[],uint8[])", "verifierHash()", "version()"};

struct Selectors {
    uint8_t risc0[R0_COUNT][4], sp1[SP1_COUNT][4];
    Selectors() {
        for (int i = 0; i < R0_COUNT; i++) fn_selector(RISC0_SIGNATURES[i], risc0[i]);
        for (int i = 0; i < SP1_COUNT; i++) fn_selector(SP1_SIGNATURES[i], sp1[i]);
    }
};
inline const Selectors& selectors() { static const Selectors s; return s; }

// ---- ABI words
inline void abi_word_u32(uint8_t* p, uint64_t v) {
    memset(p, 0, 32);
    for (int k = 0; k < 8; k++) p[31 - k] = (uint8_t)(v >> (8 * k));
}
inline void abi_word_left(uint8_t* p, const uint8_t* b, size_t n) { memset(p, 0, 32); memcpy(p, b, n); }
inline size_t abi_u8_array(uint8_t* p, const uint8_t* b, size_t n) {
    abi_word_u32(p, n);
    for (size_t i = 0; i < n; i++) abi_word_u32(p + 32 * (i + 1), b[i]);
    return 32 * (n + 1);
}

}  // namespace host
}  // namespace zkv
