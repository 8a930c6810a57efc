// Sharded (multi-device) contexts through the plain C ABI (include/zkv.h), the way a Rust / C++ host would use them:
//   zkv_risc0_ctx_create_multi(control_root, bn254_control_id, device_mask = 1)           one shard on device 0
//   zkv_ctx_create_sharded({ctx on device 0, ctx on device 0})                              two logical shards on one GPU
// argv: control_root bn254_control_id seal image_id journal_digest (hex) n.  Prints key=value pairs.  TEST ONLY.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/zkv.h"

static std::vector<uint8_t> unhex(const char* h) {
    std::vector<uint8_t> o(strlen(h) / 2);
    for (size_t i = 0; i < o.size(); i++) { unsigned v; sscanf(h + 2 * i, "%2x", &v); o[i] = (uint8_t)v; }
    return o;
}

int main(int argc, char** argv) {
    if (argc < 7) return 2;
    auto cr = unhex(argv[1]), cid = unhex(argv[2]), seal = unhex(argv[3]), id = unhex(argv[4]), jd = unhex(argv[5]);
    const size_t n = (size_t)strtoull(argv[6], nullptr, 10);
    // creation rules (no device needed)
    zkv_ctx* a = zkv_risc0_ctx_create(cr.data(), cid.data(), 0);
    zkv_ctx* b = zkv_risc0_ctx_create(cr.data(), cid.data(), 0);
    zkv_ctx* other = zkv_sp1_ctx_create(0);
    zkv_ctx* fresh = zkv_risc0_ctx_new(0);
    std::vector<uint8_t> cr2 = cr; cr2[0] ^= 1;
    zkv_ctx* diff = zkv_risc0_ctx_create(cr2.data(), cid.data(), 0);
    zkv_ctx* mix[2] = {a, other}; zkv_ctx* dup[2] = {a, a}; zkv_ctx* unin[2] = {a, fresh}; zkv_ctx* dif[2] = {a, diff};
    printf("refuse_mixed_kinds=%d refuse_duplicate=%d refuse_uninitialised=%d refuse_different_params=%d refuse_empty=%d ",
           zkv_ctx_create_sharded(mix, 2) == nullptr, zkv_ctx_create_sharded(dup, 2) == nullptr, zkv_ctx_create_sharded(unin, 2) == nullptr,
           zkv_ctx_create_sharded(dif, 2) == nullptr, zkv_ctx_create_sharded(mix, 0) == nullptr);
    zkv_ctx* two[2] = {a, b};
    zkv_ctx* sh = zkv_ctx_create_sharded(two, 2);
    zkv_ctx* multi = zkv_risc0_ctx_create_multi(cr.data(), cid.data(), 1);
    printf("sharded=%d shards=%zu dev1=%d multi_shards=%zu no_mask=%d plain_shards=%zu ", sh != nullptr, zkv_ctx_shard_count(sh), zkv_ctx_shard_device(sh, 1),
           zkv_ctx_shard_count(multi), zkv_risc0_ctx_create_multi(cr.data(), cid.data(), 0) == nullptr, zkv_ctx_shard_count(other));
    // peer access of a shard: 2 ("not applicable") until it has staged rows from another GPU; an index past the shards or a plain context is an error
    printf("peer0=%d peer1=%d peer_bad_index=%d peer_plain=%d ", zkv_ctx_shard_peer_access(sh, 0), zkv_ctx_shard_peer_access(sh, 1),
           zkv_ctx_shard_peer_access(sh, 2) < 0, zkv_ctx_shard_peer_access(other, 0) < 0);
    uint8_t sel[4] = {0, 0, 0, 0};
    zkv_risc0_get_selector(sh, sel);
    printf("selector=%02x%02x%02x%02x initialized=%d ", sel[0], sel[1], sel[2], sel[3], zkv_risc0_is_initialized(sh));
    // a batch of n copies of the real proof, every 7th with a flipped journal digest, proof 3 one byte short
    std::vector<uint8_t> blob, ids, jds; std::vector<uint64_t> off(1, 0);
    for (size_t i = 0; i < n; i++) {
        size_t len = seal.size() - (i == 3 ? 1 : 0);
        blob.insert(blob.end(), seal.begin(), seal.begin() + len); off.push_back(blob.size());
        ids.insert(ids.end(), id.begin(), id.end());
        jds.insert(jds.end(), jd.begin(), jd.end());
        if (i % 7 == 6) jds[32 * i] ^= 1;
    }
    blob.push_back(0);
    std::vector<uint8_t> st(n, 99), st1(n, 99), stm(n, 99);
    int rc = zkv_risc0_verify_batch(sh, n, blob.data(), off.data(), ids.data(), jds.data(), st.data(), nullptr);
    if (zkv_device_count() == 0) { printf("rc_no_device=%d\n", rc); return 0; }
    zkv_ctx* single = zkv_risc0_ctx_create(cr.data(), cid.data(), 0);
    int rc1 = zkv_risc0_verify_batch(single, n, blob.data(), off.data(), ids.data(), jds.data(), st1.data(), nullptr);
    int rcm = zkv_risc0_verify_batch(multi, n, blob.data(), off.data(), ids.data(), jds.data(), stm.data(), nullptr);
    size_t ok = 0, want_ok = 0, same = 0;
    for (size_t i = 0; i < n; i++) { ok += st[i] == 0; want_ok += (i != 3 && i % 7 != 6); same += st[i] == st1[i] && st[i] == stm[i]; }
    uint8_t s1 = 99, rv[4];
    int rcs = zkv_risc0_verify(sh, seal.data(), seal.size(), id.data(), jd.data(), &s1, rv);
    printf("rc=%d rc_single=%d rc_multi=%d ok=%zu want_ok=%zu same=%zu n=%zu status3=%d single_proof_rc=%d single_proof_status=%d sync=%d\n", rc, rc1, rcm, ok, want_ok,
           same, n, n > 3 ? (int)st[3] : -1, rcs, (int)s1, zkv_ctx_synchronize(sh));
    zkv_ctx_destroy(single); zkv_ctx_destroy(sh); zkv_ctx_destroy(multi); zkv_ctx_destroy(other); zkv_ctx_destroy(fresh); zkv_ctx_destroy(diff);
    return 0;
}
