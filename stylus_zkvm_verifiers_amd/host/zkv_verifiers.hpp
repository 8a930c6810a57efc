// C++ host-side mirror of the reference traits over the C ABI (include/zkv.h).
//
//   zkv::RiscZeroVerifier  <->  trait IRiscZeroVerifier + struct RiscZeroVerifier
//                               (/root/reference/contracts/src/risc0/verifier.rs:18-52)
//   zkv::Sp1Verifier       <->  trait ISp1Verifier + struct Sp1Verifier (contracts/src/sp1/verifier.rs:16-33)
//
// Same method names, argument meaning and error behaviour: the reference returns Result<_, Vec<u8>> whose error
// is the ABI-encoded Solidity custom error; here `Result::err` holds exactly those bytes.  Library/runtime failures
// (no device, HIP errors) are thrown as zkv::RuntimeError -- they are never verification outcomes.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/zkv.h"

namespace zkv {

using Bytes = std::vector<uint8_t>;
using B256 = std::array<uint8_t, 32>;
using B128 = std::array<uint8_t, 16>;
using Bytes4 = std::array<uint8_t, 4>;

struct RuntimeError : std::runtime_error {
    int code;
    RuntimeError(int c, const char* where) : std::runtime_error(std::string(where) + " failed with ZKV error " + std::to_string(c)), code(c) {}
};
inline void check(int rc, const char* where) { if (rc != ZKV_OK) throw RuntimeError(rc, where); }

template <class T>
struct Result {            // Result<T, Vec<u8>>
    bool ok;
    T value;
    Bytes err;             // ABI-encoded revert bytes when !ok
    uint8_t status;        // ZKV_STATUS_*
};

inline Bytes revert_bytes(int vm, uint8_t status, const uint8_t received[4], const uint8_t expected[4]) {
    uint8_t buf[68];
    int n = zkv_status_abi_encode(vm, status, received, expected, buf);
    if (n < 0) throw RuntimeError(n, "zkv_status_abi_encode");
    return Bytes(buf, buf + n);
}

class RiscZeroVerifier {
public:
    explicit RiscZeroVerifier(int device = 0) : ctx_(zkv_risc0_ctx_new(device)) { if (!ctx_) throw std::bad_alloc(); }
    // An `initialize`d verifier over the GPUs named by device_mask (bit d = HIP device d): every method below works unchanged, batches
    // are split over the GPUs (include/zkv.h, "sharded contexts"; SURVEY 8b `device_mask`).
    static RiscZeroVerifier multi(const B256& control_root, const B256& bn254_control_id, uint64_t device_mask) {
        zkv_ctx* c = zkv_risc0_ctx_create_multi(control_root.data(), bn254_control_id.data(), device_mask);
        if (!c) throw std::invalid_argument("zkv_risc0_ctx_create_multi: empty or invalid device mask");
        return RiscZeroVerifier(c);
    }
    RiscZeroVerifier(RiscZeroVerifier&& o) noexcept : ctx_(o.ctx_) { o.ctx_ = nullptr; }
    size_t shard_count() const { return zkv_ctx_shard_count(ctx_); }
    // Opt-in aggregate check of large batches (include/zkv.h: zkv_ctx_set_aggregate_check): sub_batch 0 = off, 16 / 32 / 64 proofs per shared
    // pairing check; seed32 = nullptr draws the secret from the operating system.  Statuses stay the deterministic ones.
    void set_aggregate_check(int sub_batch, const uint8_t* seed32 = nullptr) { check(zkv_ctx_set_aggregate_check(ctx_, sub_batch, seed32), "zkv_ctx_set_aggregate_check"); }
    std::pair<uint64_t, uint64_t> aggregate_counters() const { uint64_t o[2]; check(zkv_ctx_aggregate_counters(ctx_, o), "zkv_ctx_aggregate_counters"); return {o[0], o[1]}; }
    ~RiscZeroVerifier() { zkv_ctx_destroy(ctx_); }
    RiscZeroVerifier(const RiscZeroVerifier&) = delete;
    RiscZeroVerifier& operator=(const RiscZeroVerifier&) = delete;

    // fn initialize(&mut self, control_root: B256, bn254_control_id: B256) -> Result<(), Self::Error>
    Result<bool> initialize(const B256& control_root, const B256& bn254_control_id) {
        uint8_t st = 0;
        check(zkv_risc0_initialize(ctx_, control_root.data(), bn254_control_id.data(), &st), "zkv_risc0_initialize");
        return make(st, nullptr);
    }
    // fn verify(&self, seal: Vec<u8>, image_id: B256, journal_digest: B256) -> Result<bool, Self::Error>
    Result<bool> verify(const Bytes& seal, const B256& image_id, const B256& journal_digest) const {
        uint8_t st = 0, rv[4] = {0, 0, 0, 0};
        check(zkv_risc0_verify(ctx_, seal.data(), seal.size(), image_id.data(), journal_digest.data(), &st, rv), "zkv_risc0_verify");
        return make(st, rv);
    }
    // fn verify_integrity(&self, receipt_seal: Vec<u8>, receipt_claim_digest: B256) -> Result<bool, Self::Error>
    Result<bool> verify_integrity(const Bytes& receipt_seal, const B256& receipt_claim_digest) const {
        uint8_t st = 0, rv[4] = {0, 0, 0, 0};
        check(zkv_risc0_verify_integrity(ctx_, receipt_seal.data(), receipt_seal.size(), receipt_claim_digest.data(), &st, rv),
              "zkv_risc0_verify_integrity");
        return make(st, rv);
    }
    Bytes4 get_selector() const { Bytes4 o; check(zkv_risc0_get_selector(ctx_, o.data()), "get_selector"); return o; }
    std::pair<B128, B128> get_control_root() const {
        std::pair<B128, B128> o; check(zkv_risc0_get_control_root(ctx_, o.first.data(), o.second.data()), "get_control_root"); return o;
    }
    B256 get_bn254_control_id() const { B256 o; check(zkv_risc0_get_bn254_control_id(ctx_, o.data()), "get_bn254_control_id"); return o; }
    B256 get_verifier_key_digest() const { B256 o; check(zkv_risc0_get_verifier_key_digest(ctx_, o.data()), "get_verifier_key_digest"); return o; }
    bool is_initialized() const { return zkv_risc0_is_initialized(ctx_) != 0; }

    // Batch form: n ragged seals, n image ids, n journal digests -> n status bytes (+ received selectors).
    void verify_batch(size_t n, const uint8_t* seal_blob, const uint64_t* seal_off, const uint8_t* image_ids, const uint8_t* journal_digests,
                      uint8_t* status, uint8_t* recv_selector = nullptr) const {
        check(zkv_risc0_verify_batch(ctx_, n, seal_blob, seal_off, image_ids, journal_digests, status, recv_selector), "zkv_risc0_verify_batch");
    }
    zkv_ctx* raw() const { return ctx_; }

private:
    Result<bool> make(uint8_t st, const uint8_t* rv) const {
        Result<bool> r{st == ZKV_STATUS_OK, st == ZKV_STATUS_OK, {}, st};
        if (!r.ok) {
            Bytes4 exp = get_selector();
            static const uint8_t zero[4] = {0, 0, 0, 0};
            r.err = revert_bytes(ZKV_VM_RISC0, st, rv ? rv : zero, exp.data());
        }
        return r;
    }
    explicit RiscZeroVerifier(zkv_ctx* adopted) : ctx_(adopted) {}
    zkv_ctx* ctx_;
};

class Sp1Verifier {
public:
    explicit Sp1Verifier(int device = 0) : ctx_(zkv_sp1_ctx_create(device)) { if (!ctx_) throw std::bad_alloc(); }
    static Sp1Verifier multi(uint64_t device_mask) {       // one verifier over several GPUs (sharded context)
        zkv_ctx* c = zkv_sp1_ctx_create_multi(device_mask);
        if (!c) throw std::invalid_argument("zkv_sp1_ctx_create_multi: empty or invalid device mask");
        return Sp1Verifier(c);
    }
    Sp1Verifier(Sp1Verifier&& o) noexcept : ctx_(o.ctx_) { o.ctx_ = nullptr; }
    size_t shard_count() const { return zkv_ctx_shard_count(ctx_); }
    // Opt-in aggregate check of large batches (include/zkv.h: zkv_ctx_set_aggregate_check): sub_batch 0 = off, 16 / 32 / 64 proofs per shared
    // pairing check; seed32 = nullptr draws the secret from the operating system.  Statuses stay the deterministic ones.
    void set_aggregate_check(int sub_batch, const uint8_t* seed32 = nullptr) { check(zkv_ctx_set_aggregate_check(ctx_, sub_batch, seed32), "zkv_ctx_set_aggregate_check"); }
    std::pair<uint64_t, uint64_t> aggregate_counters() const { uint64_t o[2]; check(zkv_ctx_aggregate_counters(ctx_, o), "zkv_ctx_aggregate_counters"); return {o[0], o[1]}; }
    ~Sp1Verifier() { zkv_ctx_destroy(ctx_); }
    Sp1Verifier(const Sp1Verifier&) = delete;
    Sp1Verifier& operator=(const Sp1Verifier&) = delete;

    // fn verify_proof(&self, program_vkey: B256, public_values: Vec<u8>, proof_bytes: Vec<u8>) -> Result<(), Self::Error>
    Result<bool> verify_proof(const B256& program_vkey, const Bytes& public_values, const Bytes& proof_bytes) const {
        uint8_t st = 0, rv[4] = {0, 0, 0, 0};
        check(zkv_sp1_verify_proof(ctx_, program_vkey.data(), public_values.data(), public_values.size(), proof_bytes.data(), proof_bytes.size(),
                                   &st, rv), "zkv_sp1_verify_proof");
        Result<bool> r{st == ZKV_STATUS_OK, st == ZKV_STATUS_OK, {}, st};
        if (!r.ok) { B256 h = verifier_hash(); r.err = revert_bytes(ZKV_VM_SP1, st, rv, h.data()); }
        return r;
    }
    B256 verifier_hash() const { B256 o; zkv_sp1_verifier_hash(o.data()); return o; }
    std::string version() const { return zkv_sp1_version(); }
    void verify_batch(size_t n, const uint8_t* vkeys, const uint8_t* pv_blob, const uint64_t* pv_off, const uint8_t* proof_blob,
                      const uint64_t* proof_off, uint8_t* status, uint8_t* recv_selector = nullptr) const {
        check(zkv_sp1_verify_batch(ctx_, n, vkeys, pv_blob, pv_off, proof_blob, proof_off, status, recv_selector), "zkv_sp1_verify_batch");
    }
    zkv_ctx* raw() const { return ctx_; }

private:
    explicit Sp1Verifier(zkv_ctx* adopted) : ctx_(adopted) {}
    zkv_ctx* ctx_;
};


// ---- on-chain wire layer: eth_call calldata in, return / revert data out (include/zkv.h, "on-chain wire layer";
// Solidity view of the traits: examples/risc0-verifier/examples/interact.rs:31-43, examples/sp1-verifier/examples/interact.rs:11-19)
struct CallResult { bool reverted; Bytes data; uint8_t status; };

inline Bytes encode_verify_call(const Bytes& seal, const B256& image_id, const B256& journal_digest) {
    Bytes out(zkv_risc0_encode_verify_call(seal.data(), seal.size(), image_id.data(), journal_digest.data(), nullptr, 0));
    zkv_risc0_encode_verify_call(seal.data(), seal.size(), image_id.data(), journal_digest.data(), out.data(), out.size());
    return out;
}
inline Bytes encode_verify_integrity_call(const Bytes& seal, const B256& claim_digest) {
    Bytes out(zkv_risc0_encode_verify_integrity_call(seal.data(), seal.size(), claim_digest.data(), nullptr, 0));
    zkv_risc0_encode_verify_integrity_call(seal.data(), seal.size(), claim_digest.data(), out.data(), out.size());
    return out;
}
inline Bytes encode_verify_proof_call(const B256& program_vkey, const Bytes& public_values, const Bytes& proof_bytes) {
    Bytes out(zkv_sp1_encode_verify_proof_call(program_vkey.data(), public_values.data(), public_values.size(), proof_bytes.data(), proof_bytes.size(), nullptr, 0));
    zkv_sp1_encode_verify_proof_call(program_vkey.data(), public_values.data(), public_values.size(), proof_bytes.data(), proof_bytes.size(), out.data(),
                                     out.size());
    return out;
}
// n eth_calls against one verifier (zkv::RiscZeroVerifier or zkv::Sp1Verifier)
template <class Verifier>
inline std::vector<CallResult> eth_call_batch(const Verifier& v, const std::vector<Bytes>& calls) {
    const size_t n = calls.size();
    std::vector<uint64_t> off(n + 1, 0);
    for (size_t i = 0; i < n; i++) off[i + 1] = off[i] + calls[i].size();
    Bytes blob(off[n] + 1);
    for (size_t i = 0; i < n; i++) std::copy(calls[i].begin(), calls[i].end(), blob.begin() + off[i]);
    Bytes rev(n), st(n), ret(n * ZKV_RETURNDATA_STRIDE + 1);
    std::vector<uint32_t> len(n);
    int rc = zkv_ctx_vm(v.raw()) == ZKV_VM_SP1
                 ? zkv_sp1_eth_call_batch(v.raw(), n, blob.data(), off.data(), rev.data(), ret.data(), len.data(), st.data())
                 : zkv_risc0_eth_call_batch(v.raw(), n, blob.data(), off.data(), rev.data(), ret.data(), len.data(), st.data());
    check(rc, "zkv_eth_call_batch");
    std::vector<CallResult> out(n);
    for (size_t i = 0; i < n; i++)
        out[i] = CallResult{rev[i] != 0, Bytes(ret.begin() + i * ZKV_RETURNDATA_STRIDE, ret.begin() + i * ZKV_RETURNDATA_STRIDE + len[i]), st[i]};
    return out;
}

}  // namespace zkv
