/*
 * zkv.h -- C ABI of libzkv_mi355x.so: batched BN254 Groth16 verification for RISC Zero v2.1 and
 * SP1 v5.0.0 proofs on AMD MI355X (gfx950).  Drop-in boundary for the verify path of
 * gnosisguild/stylus-zkvm-verifiers; every entry point names the reference interface it replaces
 * (paths relative to /root/reference/contracts/src).
 *
 * Conventions
 *   - All buffers are caller-owned and borrowed for the duration of the call; the library writes only
 *     into the caller-provided outputs.  A context is opaque and library-owned.
 *   - The function return value is a LIBRARY/RUNTIME result (ZKV_OK or a negative ZKV_ERR_*); it is never
 *     a verification outcome.  Verification outcomes are per-proof status bytes (ZKV_STATUS_*), which
 *     reproduce the reference's `Result<_, Vec<u8>>` error classes in the reference's evaluation order.
 *   - There is no CPU fallback: compute entry points return ZKV_ERR_NO_DEVICE when no gfx950 device is usable.
 *   - A context is immutable after initialisation.  All calls on one context share its device workspace and are
 *     therefore executed one after another, also when the *_dev entry points are given different HIP streams (each
 *     call makes its stream wait for the previous call's last kernel).  Use one context per host thread / per
 *     concurrent stream for overlap.
 */
#ifndef ZKV_H
#define ZKV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKV_OK 0
#define ZKV_ERR_INVALID_ARG (-1)
#define ZKV_ERR_NO_DEVICE (-2)
#define ZKV_ERR_HIP (-3)
#define ZKV_ERR_OOM (-4)
#define ZKV_ERR_WRONG_CTX (-5)

/* per-proof status: common/errors.rs:3-27, risc0/errors.rs:8-44, sp1/errors.rs:8-43 */
#define ZKV_STATUS_OK 0                     /* Ok(true) / Ok(())                                        */
#define ZKV_STATUS_VERIFICATION_FAILED 1    /* VerificationFailed()                                     */
#define ZKV_STATUS_INVALID_INITIALIZATION 2 /* InvalidInitialization()                                  */
#define ZKV_STATUS_ALREADY_INITIALIZED 3    /* AlreadyInitialized()                                     */
#define ZKV_STATUS_INVALID_PROOF_DATA 4     /* InvalidProofData()                                       */
#define ZKV_STATUS_SELECTOR_MISMATCH 5      /* SelectorMismatch(bytes4,bytes4) / WrongVerifierSelector  */

#define ZKV_VM_RISC0 0
#define ZKV_VM_SP1 1
#define ZKV_SEAL_BYTES 260                  /* selector(4) + 8 x uint256: risc0/types.rs:7-13, sp1/types.rs:9-13 */

typedef struct zkv_ctx zkv_ctx;

/* ------------------------------------------------------------------ library */
/* number of usable gfx950 devices (0 when none; never fails) */
int zkv_device_count(void);
const char* zkv_version(void);

/* ------------------------------------------------------------------ RISC Zero verifier instance
 * Replaces the `RiscZeroVerifier` storage struct + `IRiscZeroVerifier` (risc0/verifier.rs:18-52). */

/* An un-initialised verifier bound to HIP device `device` (storage with initialized = false).
 * Device memory of a context, allocated at its first compute call: the key's tables (Groth16: 3 MB, plus the 16-bit window rows of the
 * vk_x stage -- 67 MB for RISC Zero, 134 MB for SP1, ZKV_MSM_WINDOW_BITS=8 leaves them out; SP1 PLONK: 24 MB) and a workspace that grows
 * with the largest chunk it has seen (about 4 KB per proof, chunks of at most 2^20 proofs). */
zkv_ctx* zkv_risc0_ctx_new(int device);
/* IRiscZeroVerifier::initialize (risc0/verifier.rs:58-76): *status = OK or ALREADY_INITIALIZED. */
int zkv_risc0_initialize(zkv_ctx* ctx, const uint8_t control_root[32], const uint8_t bn254_control_id[32], uint8_t* status);
/* new + initialize */
zkv_ctx* zkv_risc0_ctx_create(const uint8_t control_root[32], const uint8_t bn254_control_id[32], int device);
void zkv_ctx_destroy(zkv_ctx* ctx);

/* getters: get_selector / get_control_root / get_bn254_control_id / get_verifier_key_digest / is_initialized
 * (risc0/verifier.rs:106-124) */
int zkv_risc0_get_selector(const zkv_ctx* ctx, uint8_t out[4]);
int zkv_risc0_get_control_root(const zkv_ctx* ctx, uint8_t out_0[16], uint8_t out_1[16]);
int zkv_risc0_get_bn254_control_id(const zkv_ctx* ctx, uint8_t out[32]);
int zkv_risc0_get_verifier_key_digest(const zkv_ctx* ctx, uint8_t out[32]);
int zkv_risc0_is_initialized(const zkv_ctx* ctx);

/* IRiscZeroVerifier::verify (risc0/verifier.rs:78-92) over a batch.
 * seal i = seal_blob[seal_off[i] .. seal_off[i+1]); image_ids / journal_digests are n x 32 bytes.
 * status[n] receives ZKV_STATUS_*; recv_selector (n x 4, may be NULL) receives the selector found in the seal
 * for SELECTOR_MISMATCH entries (zero otherwise). */
int zkv_risc0_verify_batch(zkv_ctx* ctx, size_t n, const uint8_t* seal_blob, const uint64_t* seal_off,
                           const uint8_t* image_ids, const uint8_t* journal_digests, uint8_t* status, uint8_t* recv_selector);
/* IRiscZeroVerifier::verify_integrity (risc0/verifier.rs:94-104) over a batch. */
int zkv_risc0_verify_integrity_batch(zkv_ctx* ctx, size_t n, const uint8_t* seal_blob, const uint64_t* seal_off,
                                     const uint8_t* claim_digests, uint8_t* status, uint8_t* recv_selector);
/* Single-proof wrappers with the exact trait shapes. */
int zkv_risc0_verify(zkv_ctx* ctx, const uint8_t* seal, size_t seal_len, const uint8_t image_id[32],
                     const uint8_t journal_digest[32], uint8_t* status, uint8_t recv_selector[4]);
int zkv_risc0_verify_integrity(zkv_ctx* ctx, const uint8_t* seal, size_t seal_len, const uint8_t claim_digest[32],
                               uint8_t* status, uint8_t recv_selector[4]);
/* Fast path: fixed-stride 260-byte seals, all inputs and outputs ALREADY RESIDENT IN HBM (device pointers);
 * `stream` is a hipStream_t (NULL = the context's own stream).  Asynchronous: returns after enqueueing. */
int zkv_risc0_verify_batch_dev(zkv_ctx* ctx, size_t n, const uint8_t* d_seals, const uint8_t* d_image_ids,
                               const uint8_t* d_journal_digests, uint8_t* d_status, uint8_t* d_recv_selector, void* stream);

/* ------------------------------------------------------------------ RISC Zero verifier sets
 * Many `RiscZeroVerifier` instances -- one per (control_root, bn254_control_id), i.e. per zkVM release -- resident on one
 * device: they share the verification key (risc0/crypto.rs:16-89), hence the line tables, the e(alpha, beta) value, the
 * window tables and the workspace; per instance the device keeps only what `initialize` derives (risc0/verifier.rs:58-76,
 * 128-144): the selector (its SHA-256 chain runs in the set-up kernel), the control-id range flag and the instance's
 * share of vk_x.  Every proof of a batch names its instance; its outcome is exactly that instance's `verify` outcome.
 * An index >= the set size behaves like an un-initialised verifier (ZKV_STATUS_INVALID_INITIALIZATION). */
#define ZKV_VM_RISC0_SET 4
zkv_ctx* zkv_risc0_set_create(size_t n_instances, const uint8_t* control_roots /* n x 32 */, const uint8_t* bn254_control_ids /* n x 32 */,
                              int device);
size_t zkv_risc0_set_size(const zkv_ctx* ctx);
/* get_selector of one instance (derived on the device: needs a gfx950 device) */
int zkv_risc0_set_get_selector(zkv_ctx* ctx, size_t instance, uint8_t out[4]);
/* IRiscZeroVerifier::verify over a batch, proof i against instance[i] */
int zkv_risc0_set_verify_batch(zkv_ctx* ctx, size_t n, const uint32_t* instance, const uint8_t* seal_blob, const uint64_t* seal_off,
                               const uint8_t* image_ids, const uint8_t* journal_digests, uint8_t* status, uint8_t* recv_selector);
/* fixed-stride 260-byte seals, everything (including the instance indices) resident in HBM; asynchronous on `stream` */
int zkv_risc0_set_verify_batch_dev(zkv_ctx* ctx, size_t n, const uint32_t* d_instance, const uint8_t* d_seals, const uint8_t* d_image_ids,
                                   const uint8_t* d_journal_digests, uint8_t* d_status, uint8_t* d_recv_selector, void* stream);
/* compute_vk_x (common/groth16.rs:51-58) of (instance[i], claim digest halves i): var_signals n x 2 x 32 bytes, out n x 64 */
int zkv_risc0_set_vk_x_batch(zkv_ctx* ctx, size_t n, const uint32_t* instance, const uint8_t* var_signals, uint8_t* out);

/* ------------------------------------------------------------------ SP1 verifier
 * Replaces `Sp1Verifier` + `ISp1Verifier` (sp1/verifier.rs:16-56). */
zkv_ctx* zkv_sp1_ctx_create(int device);
int zkv_sp1_verifier_hash(uint8_t out[32]);           /* ISp1Verifier::verifier_hash, sp1/config.rs:4-9 */
const char* zkv_sp1_version(void);                    /* ISp1Verifier::version, sp1/config.rs:3          */
/* ISp1Verifier::verify_proof (sp1/verifier.rs:39-46, 58-111) over a batch: program vkeys n x 32 bytes,
 * ragged public values and proofs. */
int zkv_sp1_verify_batch(zkv_ctx* ctx, size_t n, const uint8_t* program_vkeys, const uint8_t* pv_blob, const uint64_t* pv_off,
                         const uint8_t* proof_blob, const uint64_t* proof_off, uint8_t* status, uint8_t* recv_selector);
int zkv_sp1_verify_proof(zkv_ctx* ctx, const uint8_t program_vkey[32], const uint8_t* public_values, size_t pv_len,
                         const uint8_t* proof, size_t proof_len, uint8_t* status, uint8_t recv_selector[4]);
/* Fast path, device-resident: fixed-stride 260-byte proofs and fixed-length public values. */
int zkv_sp1_verify_batch_dev(zkv_ctx* ctx, size_t n, const uint8_t* d_program_vkeys, const uint8_t* d_public_values, size_t pv_len,
                             const uint8_t* d_proofs, uint8_t* d_status, uint8_t* d_recv_selector, void* stream);

/* ------------------------------------------------------------------ mixed batches: one VM tag per proof
 * The reference's shared core takes the VM per call -- `VMType { Risc0, Sp1 }` (common/types.rs:24-26) selects the A negation and
 * the key convention in Groth16Verifier::verify_proof_with_key / verify_pairing (common/groth16.rs:23-31, 96-103) -- and a node
 * serving both deployed verifiers sees their calls interleaved (BASELINE.json config 4).  A mixed context is one RISC Zero
 * verifier (`initialize`d with the given parameters) and one SP1 verifier behind a per-proof tag: vm[i] = ZKV_VM_RISC0 means
 * proof i is `IRiscZeroVerifier::verify(seal, in_a = image_id, in_b = journal_digest)` (risc0/verifier.rs:78-92), ZKV_VM_SP1 means
 * `ISp1Verifier::verify_proof(in_a = program_vkey, in_b = public_values, seal = proof_bytes)` (sp1/verifier.rs:39-46); status and
 * received selector are exactly that verifier's.  The batch is demultiplexed on the device into two homogeneous sub-batches
 * (stable partition), which take the ordinary stage pipelines; statuses return in the caller's order.  A tag that is not a
 * VMType gets ZKV_STATUS_UNKNOWN_VM (no reference counterpart: the Rust enum cannot hold such a value). */
#define ZKV_VM_MIXED 5
#define ZKV_STATUS_UNKNOWN_VM 7
zkv_ctx* zkv_mixed_ctx_create(const uint8_t control_root[32], const uint8_t bn254_control_id[32], int device);
/* the two verifiers behind the tag (owned by the mixed context; e.g. for the getters) */
zkv_ctx* zkv_mixed_ctx_risc0(zkv_ctx* ctx);
zkv_ctx* zkv_mixed_ctx_sp1(zkv_ctx* ctx);
/* Host buffers, ragged: seal i = seal_blob[seal_off[i] .. seal_off[i+1]), in_a n x 32 bytes, in_b i = in_b_blob[in_b_off[i] ..
 * in_b_off[i+1]) (exactly 32 bytes for RISC Zero proofs, else ZKV_ERR_INVALID_ARG; any length for SP1 public values). */
int zkv_mixed_verify_batch(zkv_ctx* ctx, size_t n, const uint8_t* vm, const uint8_t* seal_blob, const uint64_t* seal_off, const uint8_t* in_a,
                           const uint8_t* in_b_blob, const uint64_t* in_b_off, uint8_t* status, uint8_t* recv_selector);
/* Fast path, everything resident in HBM: d_vm n tags, d_seals n x 260, d_in_a n x 32, d_in_b n rows of b_stride >= 32 bytes (RISC Zero
 * rows use the first 32 bytes, SP1 rows the first pv_len <= b_stride bytes).  Asynchronous on `stream` after one synchronisation
 * (the host learns the two sub-batch sizes from the device). */
int zkv_mixed_verify_batch_dev(zkv_ctx* ctx, size_t n, const uint8_t* d_vm, const uint8_t* d_seals, const uint8_t* d_in_a, const uint8_t* d_in_b,
                               size_t b_stride, size_t pv_len, uint8_t* d_status, uint8_t* d_recv_selector, void* stream);

/* ------------------------------------------------------------------ sharded (multi-device) contexts
 * `zkv_risc0_ctx_create(control_root, bn254_control_id, device_mask)` -- one verifier over several GPUs of a node.
 * A sharded context is a set of single-device contexts of ONE verifier (same kind, same parameters) behind the ordinary batch entry
 * points: proofs are independent (the reference verifies one per call: risc0/verifier.rs:78-92, sp1/verifier.rs:39-46), so a batch is
 * split into contiguous ranges, one per shard, and nothing is exchanged between shards; statuses land in the caller's order.
 *   - host-buffer batches (zkv_*_verify_batch, zkv_groth16_verify_batch, zkv_*_eth_call_batch): one host thread per shard runs the
 *     single-device entry point on its range of the caller's buffers -- every GPU pulls its rows over its own PCIe link;
 *   - device-resident batches (zkv_*_verify_batch_dev): the rows may live on any GPU; each shard on another GPU receives its range with
 *     hipMemcpyPeerAsync (one direct xGMI link per peer) in two pieces, the second behind the first piece's kernels, and copies its
 *     statuses back the same way.  With a non-NULL `stream` (a stream of the GPU holding the rows) the shards start after what that
 *     stream has enqueued and the stream continues after all statuses are back; with NULL use zkv_ctx_synchronize.
 * Shards with fewer than ZKV_SHARD_MIN proofs (environment, default 1,024) are not used: a single proof runs on shard 0.  Getters
 * answer for the common parameters; entry points that are not listed above run on shard 0.  The function return value stays a
 * library / runtime result (the first failing shard's). */
/* Takes ownership of the contexts in `shards` on success (destroy only the returned context).  Their kind must be one of RISC0, SP1,
 * MIXED, GROTH16, SP1_PLONK, identical across shards, initialised, with identical parameters; a device may carry several shards.
 * NULL on any violation (the shards are then still the caller's). */
zkv_ctx* zkv_ctx_create_sharded(zkv_ctx* const* shards, size_t n_shards);
size_t zkv_ctx_shard_count(const zkv_ctx* ctx);       /* 0 for a single-device context */
int zkv_ctx_shard_device(const zkv_ctx* ctx, size_t shard);
/* One shard per set bit of device_mask (bit d = HIP device d). */
zkv_ctx* zkv_risc0_ctx_create_multi(const uint8_t control_root[32], const uint8_t bn254_control_id[32], uint64_t device_mask);
zkv_ctx* zkv_sp1_ctx_create_multi(uint64_t device_mask);
zkv_ctx* zkv_mixed_ctx_create_multi(const uint8_t control_root[32], const uint8_t bn254_control_id[32], uint64_t device_mask);

/* ------------------------------------------------------------------ SP1 PLONK verifier (BASELINE.json configs[4])
 * `ISp1Verifier::verify_proof` (sp1/verifier.rs:16-29, 39-46, 58-111) with the PLONK proof system behind it -- the path the reference
 * marks "in progress" (README.md:25, contracts/src/lib.rs:11) and for which it holds no code, key or proof: PARITY UNPINNED BY
 * CONSTRUCTION.  The algorithm is gnark's BN254 PLONK verifier (v0.10-0.11: SHA-256 Fiat-Shamir transcript, one BSB22 commitment,
 * linearised polynomial with the quotient folded in, batched KZG opening, one 2-pair pairing), restated in oracle/plonk_model.py;
 * proof_bytes = 4-byte selector (first bytes of `verifier_hash`) + 27 words = ZKV_PLONK_PROOF_BYTES.  Check order and statuses are
 * those of the Groth16 `verify_proof`: INVALID_PROOF_DATA (length < 4), SELECTOR_MISMATCH, INVALID_PROOF_DATA (length != 868),
 * VERIFICATION_FAILED (program_vkey >= R, a scalar >= R, a point off the curve, the algebraic relation or the pairing failing), OK.
 * The verifying key is supplied by the caller (no SP1 PLONK key exists in the reference): 32-byte big-endian words
 *   size | size_inv | generator | coset_shift | nb_public (= 2) | n_qcp (= 1: SP1's circuit has one BSB22 commitment) | commitment_constraint_index |
 *   S1 S2 S3 Ql Qr Qm Qo Qk Qcp (G1 x, y) | G2 generator | [tau]G2 (EIP-197 order x_im x_re y_im y_re)
 * (1,056 bytes; any other shape returns NULL).  A key holding an invalid point fails every proof. */
#define ZKV_VM_SP1_PLONK 6
#define ZKV_PLONK_PROOF_BYTES 868
zkv_ctx* zkv_sp1_plonk_ctx_create(const uint8_t* vk_bytes, size_t vk_len, const uint8_t verifier_hash[32], int device);
int zkv_sp1_plonk_verifier_hash(const zkv_ctx* ctx, uint8_t out[32]);                       /* ISp1Verifier::verifier_hash */
int zkv_sp1_plonk_verify_batch(zkv_ctx* ctx, size_t n, const uint8_t* program_vkeys, const uint8_t* pv_blob, const uint64_t* pv_off,
                               const uint8_t* proof_blob, const uint64_t* proof_off, uint8_t* status, uint8_t* recv_selector);
int zkv_sp1_plonk_verify_proof(zkv_ctx* ctx, const uint8_t program_vkey[32], const uint8_t* public_values, size_t pv_len,
                               const uint8_t* proof, size_t proof_len, uint8_t* status, uint8_t recv_selector[4]);
/* Fast path, device-resident: fixed-stride 868-byte proofs and fixed-length public values; asynchronous on `stream`. */
int zkv_sp1_plonk_verify_batch_dev(zkv_ctx* ctx, size_t n, const uint8_t* d_program_vkeys, const uint8_t* d_public_values, size_t pv_len,
                                   const uint8_t* d_proofs, uint8_t* d_status, uint8_t* d_recv_selector, void* stream);

/* ------------------------------------------------------------------ on-chain wire layer: eth_call batches
 * What a client of the deployed example shells sends: calldata for the Solidity view of the two traits
 * (examples/risc0-verifier/examples/interact.rs:31-43, examples/sp1-verifier/examples/interact.rs:11-19; the shells are
 * examples/{risc0,sp1}-verifier/src/lib.rs).  Stylus exports `Vec<u8>` as `uint8[]`, so every seal byte travels as one
 * 32-byte word: `verify(uint8[],bytes32,bytes32)`, `verifyIntegrity(uint8[],bytes32)`, `verifyProof(bytes32,uint8[],uint8[])`.
 * Calldata is decoded on the device.  UNPINNED (the Stylus router is not part of the reference tree): calldata must be the
 * canonical ABI encoding of its arguments (alloy-sol-types 0.8.20 `abi_decode_params(.., validate = true)`); anything else,
 * and any unknown function selector, reverts with empty return data and reports ZKV_STATUS_BAD_CALLDATA. */
#define ZKV_STATUS_BAD_CALLDATA 6           /* wire layer only: the contract's router cannot decode the call */
#define ZKV_RETURNDATA_STRIDE 96            /* longest return / revert data of any method (SP1 `version()`)   */
/* first 4 bytes of keccak-256(signature), e.g. "verify(uint8[],bytes32,bytes32)" */
int zkv_abi_function_selector(const char* signature, uint8_t out[4]);
/* Canonical calldata of one call.  Returns the length needed; writes only when out != NULL and cap is large enough. */
size_t zkv_risc0_encode_verify_call(const uint8_t* seal, size_t seal_len, const uint8_t image_id[32],
                                    const uint8_t journal_digest[32], uint8_t* out, size_t cap);
size_t zkv_risc0_encode_verify_integrity_call(const uint8_t* seal, size_t seal_len, const uint8_t claim_digest[32], uint8_t* out, size_t cap);
size_t zkv_sp1_encode_verify_proof_call(const uint8_t program_vkey[32], const uint8_t* public_values, size_t pv_len,
                                        const uint8_t* proof, size_t proof_len, uint8_t* out, size_t cap);
/* n eth_calls against one verifier instance: request i = calldata_blob[calldata_off[i] .. calldata_off[i+1]).
 * reverted[i] = 0 / 1; returndata (n x ZKV_RETURNDATA_STRIDE) + returndata_len[n] hold the ABI-encoded return value
 * (`true` word for verify / verifyIntegrity, nothing for verifyProof, the getters' values) or the revert bytes
 * (common/errors.rs:18-27, risc0/errors.rs:21-32, sp1/errors.rs:21-32).  status (may be NULL) receives ZKV_STATUS_* of
 * verify-class calls and ZKV_STATUS_BAD_CALLDATA for everything the device did not verify (getters, `initialize`, which
 * is simulated without storing anything, and undecodable calldata). */
int zkv_risc0_eth_call_batch(zkv_ctx* ctx, size_t n, const uint8_t* calldata_blob, const uint64_t* calldata_off, uint8_t* reverted,
                             uint8_t* returndata, uint32_t* returndata_len, uint8_t* status);
int zkv_sp1_eth_call_batch(zkv_ctx* ctx, size_t n, const uint8_t* calldata_blob, const uint64_t* calldata_off, uint8_t* reverted,
                           uint8_t* returndata, uint32_t* returndata_len, uint8_t* status);
/* Fast path: calldata blob (calldata_bytes long) and its n+1 offsets ALREADY RESIDENT IN HBM; handles the verify-class
 * calls of the context's verifier (anything else gets ZKV_STATUS_BAD_CALLDATA -- resolve those on the host).
 * Asynchronous on `stream` (NULL = the context's stream). */
int zkv_eth_call_batch_dev(zkv_ctx* ctx, size_t n, const uint8_t* d_calldata, const uint64_t* d_calldata_off, uint64_t calldata_bytes,
                           uint8_t* d_status, uint8_t* d_recv_selector, void* stream);
/* return / revert data of one verify-class call from its status byte (companion of the device fast path) */
int zkv_eth_call_returndata(const zkv_ctx* ctx, uint8_t status, const uint8_t recv_selector[4], uint8_t out[ZKV_RETURNDATA_STRIDE],
                            uint32_t* out_len, uint8_t* reverted);
/* HIP-event duration (ms) of the calldata-decode kernel of the most recent eth_call chunk. */
int zkv_ctx_last_wire_ms(zkv_ctx* ctx, float* out_ms);

/* ------------------------------------------------------------------ precompile-level batches (the inner seam)
 * The three EVM precompiles the reference STATICCALLs (common/groth16.rs:12-14): ecAdd 0x06 (call site :55),
 * ecMul 0x07 (:54), ecPairing 0x08 (:121-125), with EIP-196/197 semantics.  ok[i] = 1 when call i succeeds, 0 when
 * the precompile would fail (coordinate >= Q, point off curve / off twist / outside the order-r subgroup): the
 * reference maps that to Err(()) (groth16.rs:60-73, 109-128).  Host buffers. */
#define ZKV_VM_BN254 2
zkv_ctx* zkv_bn254_ctx_create(int device);
/* in: n x 128 bytes (x1 y1 x2 y2), out: n x 64 bytes */
int zkv_bn254_ecadd_batch(zkv_ctx* ctx, size_t n, const uint8_t* in, uint8_t* out, uint8_t* ok);
/* in: n x 96 bytes (x y scalar), out: n x 64 bytes */
int zkv_bn254_ecmul_batch(zkv_ctx* ctx, size_t n, const uint8_t* in, uint8_t* out, uint8_t* ok);
/* in: n calls of k pairs each, k x 192 bytes per call (G1 x y, G2 x_im x_re y_im y_re); result[i] = 1 iff the product
 * of the k pairings is 1 (the precompile's 32-byte output word), meaningful when ok[i] = 1. */
int zkv_bn254_pairing_batch(zkv_ctx* ctx, size_t n, size_t k, const uint8_t* in, uint8_t* result, uint8_t* ok);
/* The same with calldata, results and verdicts resident in HBM (device pointers), enqueued on `stream` (a hipStream_t; NULL = the context's
 * stream) without copies or synchronisation: zkv_ctx_synchronize or the stream tells when result / ok are written. */
int zkv_bn254_pairing_batch_dev(zkv_ctx* ctx, size_t n, size_t k, const uint8_t* d_in, uint8_t* d_result, uint8_t* d_ok, void* stream);

/* ------------------------------------------------------------------ Groth16 core, arbitrary verification key
 * Groth16Verifier::verify_proof_with_key(vm_type, &vk, a, b, c, &signals) -> bool (common/groth16.rs:23-49) is generic over the
 * key; this context takes any key.  vk_words: alpha1.x alpha1.y | beta2.x[0] x[1] y[0] y[1] | gamma2 (4) | delta2 (4) |
 * ic[0].x ic[0].y ... ic[n_ic-1] -- 32-byte big-endian words in the layout of `VerificationKey` (common/types.rs:17-23; G2 words
 * are (imaginary, real)), 1 <= n_ic <= 6.  vm_type: ZKV_VM_RISC0 negates A, ZKV_VM_SP1 does not (groth16.rs:96-103).
 * A key holding a point the precompiles would reject makes every proof fail, as in the reference. */
#define ZKV_VM_GROTH16 3
zkv_ctx* zkv_groth16_ctx_create(const uint8_t* vk_words, size_t n_ic, int vm_type, int device);
/* proofs: n x 256 bytes (a.x a.y b.x[0] b.x[1] b.y[0] b.y[1] c.x c.y); signals: n x (n_ic - 1) x 32 bytes big-endian;
 * verified[i] = 1 / 0 is the function's return value (signal >= R, malformed point, pairing product != 1 -> 0). */
int zkv_groth16_verify_batch(zkv_ctx* ctx, size_t n, const uint8_t* proofs, const uint8_t* signals, uint8_t* verified);

/* ------------------------------------------------------------------ Groth16 core pieces
 * Groth16Verifier::compute_vk_x (common/groth16.rs:51-58) for a batch: vk_x = IC[0] + sum s_i IC[i+1] with the context's
 * fixed signals (RISC Zero: control root halves and bn254 control id) and the per-proof signals given here as
 * n x k x 32 big-endian bytes: k = 2 for a RISC Zero (claim digest low / high halves) or SP1 (program vkey, public-values hash)
 * context, k = n_ic - 1 (all signals) for a ZKV_VM_GROTH16 context.
 * Signals must be < R (the reference rejects the proof before this step otherwise).  out: n x 64 bytes (x, y), (0,0) = infinity. */
int zkv_ctx_vk_x_batch(zkv_ctx* ctx, size_t n, const uint8_t* var_signals, uint8_t* out);

/* ------------------------------------------------------------------ diagnostics
 * Secondary roofline (no reference counterpart): rate of the library's own register-resident Montgomery multiplication on this
 * device, in multiplications per second over the whole chip, measured in time (HIP events).  kind 0 = fp_mul (one product + one
 * reduction), kind 1 = the lane-pair Fp2 product (two products + one reduction per lane, counted as two multiplications -- the
 * unit in which the verify kernels' work is counted); waves_per_simd 1..8 resident wavefronts per SIMD; iters loop trips (four
 * calls each).  *shader_clock_ghz (may be NULL) = shader clock under this load from s_memtime / s_memrealtime. */
int zkv_diag_mulmod_rate(int device, int kind, int waves_per_simd, uint32_t iters, double* mulmods_per_s, double* shader_clock_ghz);
/* Independent issue-rate roof: lane-instructions per second over the whole chip of nothing but one instruction on register-resident
 * operands (eight independent chains per lane, 64 instructions per loop trip).  kind 0 = v_mad_u64_u32 (the 32 x 32 + 64 multiply-add
 * every field multiplication of the verify kernels is made of), kind 1 = v_mad_i64_i32, kind 2 = v_add_u32 (a plain VOP2 instruction,
 * for scale).  Unlike zkv_diag_mulmod_rate this roof does not move when the library's own multiplier changes. */
int zkv_diag_issue_rate(int device, int kind, int waves_per_simd, uint32_t iters, double* lane_instr_per_s, double* shader_clock_ghz);

/* ------------------------------------------------------------------ shared */
int zkv_ctx_vm(const zkv_ctx* ctx);                   /* ZKV_VM_* */
/* Tuning knob (no reference counterpart): kernel mapping of the G2 / Miller / final-exponentiation stages.
 * 0 = automatic (default): one proof per pair of lanes; for chunks of at most ZKV_WIDE_BELOW proofs (environment, default 12288)
 * one proof per 16 lanes, which halves the latency of a small batch; for chunks of at most ZKV_WAVE_BELOW proofs (default 2048)
 * one proof per WAVEFRONT (64 lanes), the lowest latency -- the case of the reference's own API, one proof per call
 * (risc0/verifier.rs:78-92) -- and for chunks of at most ZKV_DUAL_BELOW proofs (default 768) the Miller loop gets a second
 * wavefront per proof, which steps the running G2 point ahead of the accumulator.  2 = always lane pairs; 16 = always 16 lanes per
 * proof; 64 = always one proof per wavefront; 128 = always two wavefronts per proof in the Miller loop.  Results are identical.  (The round-1 one-proof-per-lane kernels were retired: 2.53 against 3.11 M proofs/s.) */
int zkv_ctx_set_lanes_per_proof(zkv_ctx* ctx, int lanes);
/* Aggregate check (no reference counterpart; off by default).  The reference answers one proof per call with one pairing check
 * (common/groth16.rs:60-72, 109-128).  A batch may share that check: with enable != 0, chunks of at least ZKV_AGG_MIN proofs
 * (environment, default 131072) are checked in sub-batches of 16, 32, 64, 128 or 256 proofs (enable = that size: small sub-batches
 * mean more shared checks but fewer proofs verified again when one fails; enable = 1: automatic -- 32 at first, then, whenever the
 * context is idle at the start of a chunk, the size that suits the failure rate the counters show) through ONE product of
 * pairings per sub-batch,
 *     prod_i e(r_i (-A_i), B_i) * e(sum_i r_i vk_x_i, gamma) * e(sum_i r_i C_i, delta) * e((sum_i r_i) alpha, beta) == 1,
 * with 128-bit coefficients r_i derived (SHA-256) from 32 secret bytes and a per-chunk counter.  Every check before the pairing
 * equation stays per proof and deterministic (seal format, selector, signal ranges, curve membership of A and C, curve and subgroup
 * membership of B).  A sub-batch whose aggregate check fails is verified again proof by proof by the ordinary kernels, so the
 * statuses are the deterministic ones unless invalid proofs pass an aggregate check, which happens with probability 2^-128 per
 * sub-batch over the coefficients -- provided the proofs were fixed before the secret was drawn.  seed32 = NULL draws the secret from
 * the operating system (getrandom) -- the setting for production: the secret is then drawn afresh every 1,024 chunks; a
 * caller-supplied seed makes runs reproducible (tests), must not be known to whoever supplies proofs and must never be used for a
 * second context or process (the per-chunk counter restarts at zero: the same seed replays the same coefficients).  In the automatic
 * mode the check also switches itself OFF for a while -- 8, then 16 ... 64 chunks, probing again in between -- while more than one
 * sub-batch of 16 in three fails, where it would cost more than it saves.  Applies to RISC Zero, SP1 (Groth16), verifier-set, generic-key, mixed and sharded
 * contexts, and to SP1 PLONK contexts (both pairs of a PLONK check are fixed: prod_i (e(D_i, [1]_2) e(-Q_i, [tau]_2))^{r_i} needs two
 * scalar multiplications per proof and one pairing product per sub-batch -- no per-proof Miller loop is left); ZKV_ERR_INVALID_ARG on
 * a precompile context.  Keys with alpha or beta at infinity fall back to the ordinary path, and so does a context whose extra buffers
 * (about 0.9 KB of HBM per proof in flight on top of the workspace's 3.7 KB) cannot be allocated -- zkv_ctx_aggregate_counters shows
 * whether chunks were checked in aggregate.  zkv_ctx_last_stage_ms then reports: [1] the per-proof G1 scalar multiplications, [3] the
 * Miller loops (variable pairs, sub-batch sums, pseudo-proofs), [4] everything after (pseudo-proofs' final exponentiation, second pass).
 * Throughput: see DESIGN.md (2^20 SP1 proofs: 12.0 M proofs/s all valid with sub-batches of 128, 9.6 M with one proof in 64 rejected and sub-batches of 16, against 5.7 M; a proof
 * rejected at the pairing costs its sub-batch a second, ordinary pass, and small chunks gain nothing). */
int zkv_ctx_set_aggregate_check(zkv_ctx* ctx, int enable, const uint8_t* seed32);
/* out[0] = sub-batches checked in aggregate, out[1] = those that failed and were verified proof by proof, since device set-up.
 * Synchronise (zkv_ctx_synchronize or the stream) with the batches to be counted first. */
int zkv_ctx_aggregate_counters(zkv_ctx* ctx, uint64_t out[2]);
/* Device set-up (stream, verification-key tables) and per-chunk buffers for batches of up to n proofs, ahead of the first
 * batch call.  Optional: every batch entry point does this on demand; the buffers (about 3.7 KB per proof in flight) grow to the
 * largest batch seen, at most ZKV_CHUNK proofs (environment, default 2^20; larger batches run chunk by chunk). */
int zkv_ctx_reserve(zkv_ctx* ctx, size_t n);
/* The two-wavefront small-batch kernels (chunks of at most ZKV_DUAL_BELOW proofs; ecPairing calls in small batches) let one wavefront wait for
 * another with a BOUNDED spin (about half a second; nothing can hang).  A wait that runs out -- only a wavefront that died can cause it -- fails
 * closed: the proof / call is reported as failing, never as accepted.  So that such an event is not mistaken for a verdict, it is counted per device
 * since the library was loaded; out = that count (0 in every run so far; the tests assert it).  Synchronise first. */
int zkv_diag_wait_faults(int device, uint64_t* out);
/* Sharded context, device-resident batches: how shard `shard` reaches the GPU that held the rows of the last batch it staged --
 * 1: peer access granted (hipDeviceEnablePeerAccess: the range travels as direct xGMI copies), 0: refused (the runtime bounces the
 * copies through the host), 2: not applicable so far (the shard sits on the source GPU, or it has staged nothing yet).
 * ZKV_ERR_INVALID_ARG (negative) for a context that is not sharded or an index past its shards.  A scaling run reads this to tell direct copies
 * from bounced ones. */
int zkv_ctx_shard_peer_access(zkv_ctx* ctx, size_t shard);
/* Host buffers that a caller hands to the batch entry points again and again (a server's receive ring: seals, public inputs, the
 * status array) can be pinned once: the H2D staging of a host-buffer batch then runs as direct DMA from them instead of going through
 * the runtime's pageable-copy path, and the first segment's copy -- the only one the kernels do not hide -- shrinks accordingly
 * (SURVEY 8(d) measures the metric at this boundary: "wall-clock over the batch call at the C ABI, H2D staging included").
 * Thin wrappers of hipHostRegister / hipHostUnregister (portable across devices); the memory stays the caller's.  Optional: every entry
 * point accepts pageable memory.  Without a usable device both return ZKV_ERR_NO_DEVICE; what the runtime refuses is ZKV_ERR_HIP. */
int zkv_host_register(void* ptr, size_t bytes);
int zkv_host_unregister(void* ptr);
/* The chunk size in force: ZKV_CHUNK rounded up to a multiple of 64 and clamped to [64, 2^26] (the kernels address a chunk's
 * workspace rows through 32-bit lane offsets, which a larger chunk would wrap). */
size_t zkv_chunk_capacity(void);
/* Blocks until everything enqueued by calls on this context has finished (device-wide wait; on a mixed context this covers
 * both verifiers behind the tag, whichever of them the last batch used). */
int zkv_ctx_synchronize(zkv_ctx* ctx);
/* HIP-event durations (ms) of the stages of the most recent batch chunk on this context:
 * [0] prep (parse + SHA-256 + point validation)  [1] vk_x MSM + normalisation  [2] G2 subgroup check (a stage of its own only when
 * the 16-lane kernels run in line; the lane-pair Miller loop is the subgroup test itself, and the time is then ~0)
 * [3] Miller loop  [4] final exponentiation.  Synchronises the context.  A batch larger than one chunk (and a host-buffer batch,
 * which is staged segment by segment) reports its LAST chunk / segment only; a mixed context reports the sum over the sub-batches
 * of its most recent call (zeros for a VM that call held no proof of). */
int zkv_ctx_last_stage_ms(zkv_ctx* ctx, float out_ms[5]);
/* Revert bytes of a status exactly as the reference ABI-encodes its errors (common/errors.rs:18-27,
 * risc0/errors.rs:21-32, sp1/errors.rs:21-32).  Returns the length written (0, 4 or 68) or a negative error. */
int zkv_status_abi_encode(int vm, uint8_t status, const uint8_t received[4], const uint8_t expected[4], uint8_t out[68]);

#ifdef __cplusplus
}
#endif
#endif /* ZKV_H */
