"""Host-side mirror of `IRiscZeroVerifier` / `RiscZeroVerifier`
(/root/reference/contracts/src/risc0/verifier.rs:18-196) over the C ABI of libzkv_mi355x.so.

Same method names, argument meaning and error behaviour as the reference trait; the batch methods are the
MI355X-native addition (one call, many proofs, one status byte per proof)."""
import ctypes as C

import numpy as np

from . import _lib
from .errors import (STATUS_OK, STATUS_SELECTOR_MISMATCH, VM_RISC0, VerifierError)


def _blob(items):
    off = np.zeros(len(items) + 1, dtype=np.uint64)
    if len(items):
        off[1:] = np.cumsum([len(s) for s in items], dtype=np.uint64)
    return b''.join(bytes(s) for s in items) + b'\0', off


def _cat32(items, what):
    for x in items:
        if len(x) != 32:
            raise ValueError('%s must be 32 bytes' % what)
    return b''.join(bytes(x) for x in items) + b'\0'


def _same_len(n, **lists):
    """Every per-proof list of a batch must have one entry per proof: the C side reads n rows from each."""
    for name, v in lists.items():
        if len(v) != n:
            raise ValueError('%s has %d entries for a batch of %d proofs' % (name, len(v), n))


def _set_aggregate_check(L, h, enable, seed, sub_batch=None):
    """zkv_ctx_set_aggregate_check: seed = None draws the secret from the operating system; 32 bytes make a run reproducible.
    sub_batch = None: automatic (32 proofs at first, then what suits the failure rate seen); 16 ... 256: fixed."""
    if seed is not None and len(seed) != 32: raise ValueError('seed must be 32 bytes')
    _lib.check(L.zkv_ctx_set_aggregate_check(h, (1 if sub_batch is None else int(sub_batch)) if enable else 0, bytes(seed) if seed is not None else None),
               'zkv_ctx_set_aggregate_check')


def _aggregate_counters(L, h):
    out = (C.c_uint64 * 2)()
    _lib.check(L.zkv_ctx_aggregate_counters(h, out), 'zkv_ctx_aggregate_counters')
    return int(out[0]), int(out[1])


class RiscZeroVerifier:
    def __init__(self, device=0):
        self._L = _lib.lib()
        self._h = self._L.zkv_risc0_ctx_new(device)
        if not self._h:
            raise MemoryError('zkv_risc0_ctx_new')

    def close(self):
        if getattr(self, '_h', None):
            self._L.zkv_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- IRiscZeroVerifier
    def initialize(self, control_root, bn254_control_id):
        """verifier.rs:58-76.  Raises VerifierError(AlreadyInitialized) on a second call."""
        st = C.c_uint8(0)
        _lib.check(self._L.zkv_risc0_initialize(self._h, bytes(control_root), bytes(bn254_control_id), C.byref(st)),
                   'zkv_risc0_initialize')
        if st.value != STATUS_OK:
            raise VerifierError(VM_RISC0, st.value)

    def verify(self, seal, image_id, journal_digest):
        """verifier.rs:78-92: returns True or raises VerifierError (the reference never returns Ok(false))."""
        st = C.c_uint8(0); rv = C.create_string_buffer(4)
        _lib.check(self._L.zkv_risc0_verify(self._h, bytes(seal), len(seal), bytes(image_id), bytes(journal_digest),
                                            C.byref(st), rv), 'zkv_risc0_verify')
        return self._result(st.value, rv.raw)

    def verify_integrity(self, receipt_seal, receipt_claim_digest):
        """verifier.rs:94-104."""
        st = C.c_uint8(0); rv = C.create_string_buffer(4)
        _lib.check(self._L.zkv_risc0_verify_integrity(self._h, bytes(receipt_seal), len(receipt_seal),
                                                      bytes(receipt_claim_digest), C.byref(st), rv), 'zkv_risc0_verify_integrity')
        return self._result(st.value, rv.raw)

    def get_selector(self):
        o = C.create_string_buffer(4); _lib.check(self._L.zkv_risc0_get_selector(self._h, o), 'get_selector'); return o.raw

    def get_control_root(self):
        a, b = C.create_string_buffer(16), C.create_string_buffer(16)
        _lib.check(self._L.zkv_risc0_get_control_root(self._h, a, b), 'get_control_root')
        return a.raw, b.raw

    def get_bn254_control_id(self):
        o = C.create_string_buffer(32); _lib.check(self._L.zkv_risc0_get_bn254_control_id(self._h, o), 'get_bn254_control_id'); return o.raw

    def get_verifier_key_digest(self):
        o = C.create_string_buffer(32); _lib.check(self._L.zkv_risc0_get_verifier_key_digest(self._h, o), 'get_verifier_key_digest'); return o.raw

    def is_initialized(self):
        return bool(self._L.zkv_risc0_is_initialized(self._h))

    def _result(self, status, recv):
        if status == STATUS_OK:
            return True
        if status == STATUS_SELECTOR_MISMATCH:
            raise VerifierError(VM_RISC0, status, recv, self.get_selector())
        raise VerifierError(VM_RISC0, status)

    # ---- batch (host buffers, ragged seals)
    def verify_batch(self, seals, image_ids, journal_digests):
        """One status byte per proof (errors.STATUS_*), plus the received selector of mismatching seals."""
        n = len(seals)
        _same_len(n, image_ids=image_ids, journal_digests=journal_digests)
        blob, off = _blob(seals)
        st = np.zeros(n, dtype=np.uint8); rv = np.zeros((n, 4), dtype=np.uint8)
        _lib.check(self._L.zkv_risc0_verify_batch(self._h, n, blob, off.ctypes.data, _cat32(image_ids, 'image_id'),
                                                  _cat32(journal_digests, 'journal_digest'), st.ctypes.data, rv.ctypes.data),
                   'zkv_risc0_verify_batch')
        return st, rv

    def verify_integrity_batch(self, seals, claim_digests):
        n = len(seals)
        _same_len(n, claim_digests=claim_digests)
        blob, off = _blob(seals)
        st = np.zeros(n, dtype=np.uint8); rv = np.zeros((n, 4), dtype=np.uint8)
        _lib.check(self._L.zkv_risc0_verify_integrity_batch(self._h, n, blob, off.ctypes.data, _cat32(claim_digests, 'claim_digest'),
                                                            st.ctypes.data, rv.ctypes.data), 'zkv_risc0_verify_integrity_batch')
        return st, rv

    # ---- batch, inputs already resident in HBM (device pointers as ints; 260-byte stride)
    def verify_batch_dev(self, n, d_seals, d_image_ids, d_journal_digests, d_status, d_recv=0, stream=0):
        _lib.check(self._L.zkv_risc0_verify_batch_dev(self._h, n, d_seals, d_image_ids, d_journal_digests, d_status,
                                                      d_recv or None, stream or None), 'zkv_risc0_verify_batch_dev')

    def vk_x_batch(self, var_signals):
        """Groth16Verifier::compute_vk_x (common/groth16.rs:51-58) for a batch: var_signals = list of (s_a, s_b) 32-byte pairs
        (the two per-proof signals); returns 64-byte affine points."""
        n = len(var_signals)
        blob = b''.join(bytes(a) + bytes(b) for a, b in var_signals) + b'\0'
        out = np.zeros(max(64 * n, 1), dtype=np.uint8)
        _lib.check(self._L.zkv_ctx_vk_x_batch(self._h, n, blob, out.ctypes.data), 'zkv_ctx_vk_x_batch')
        return [out[64 * i:64 * i + 64].tobytes() for i in range(n)]

    def set_lanes_per_proof(self, lanes):
        """Kernel mapping of the Fp2-heavy stages: 0 = automatic (lane pairs; 16 lanes per proof for small chunks, one proof per
        wavefront for the smallest), 2 = lane pairs, 16 = sixteen lanes per proof, 64 = one proof per wavefront; same results."""
        _lib.check(self._L.zkv_ctx_set_lanes_per_proof(self._h, lanes), 'zkv_ctx_set_lanes_per_proof')

    def reserve(self, n):
        """Device set-up and per-chunk buffers for batches of up to n proofs, ahead of the first batch (optional)."""
        _lib.check(self._L.zkv_ctx_reserve(self._h, n), 'zkv_ctx_reserve')

    def set_aggregate_check(self, enable=True, seed=None, sub_batch=None):
        """Opt-in: share the pairing check among sub-batches of 16 ... 256 proofs (None: chosen by the failure rate seen) of a large chunk (include/zkv.h, csrc/zkv_agg.h);
        statuses stay the deterministic ones (a failed sub-batch is verified again proof by proof)."""
        _set_aggregate_check(self._L, self._h, enable, seed, sub_batch)

    def aggregate_counters(self):
        """(sub-batches checked in aggregate, sub-batches that failed and were verified proof by proof)."""
        return _aggregate_counters(self._L, self._h)

    def synchronize(self):
        _lib.check(self._L.zkv_ctx_synchronize(self._h), 'zkv_ctx_synchronize')

    def last_stage_ms(self):
        out = (C.c_float * 5)()
        _lib.check(self._L.zkv_ctx_last_stage_ms(self._h, out), 'zkv_ctx_last_stage_ms')
        return list(out)


class RiscZeroVerifierSet:
    """Many `RiscZeroVerifier` instances (one per (control_root, bn254_control_id)) resident on one device and sharing the
    verification-key tables; every proof of a batch names its instance (include/zkv.h, "RISC Zero verifier sets")."""

    def __init__(self, control_roots, bn254_control_ids, device=0):
        assert len(control_roots) == len(bn254_control_ids) and len(control_roots) > 0
        self._L = _lib.lib()
        self._h = self._L.zkv_risc0_set_create(len(control_roots), _cat32(control_roots, 'control_root'),
                                               _cat32(bn254_control_ids, 'bn254_control_id'), device)
        if not self._h:
            raise MemoryError('zkv_risc0_set_create')

    def close(self):
        if getattr(self, '_h', None):
            self._L.zkv_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def __len__(self):
        return self._L.zkv_risc0_set_size(self._h)

    def get_selector(self, instance):
        o = C.create_string_buffer(4)
        _lib.check(self._L.zkv_risc0_set_get_selector(self._h, instance, o), 'zkv_risc0_set_get_selector')
        return o.raw

    def verify_batch(self, instances, seals, image_ids, journal_digests):
        n = len(seals)
        _same_len(n, instances=instances, image_ids=image_ids, journal_digests=journal_digests)
        blob, off = _blob(seals)
        idx = np.ascontiguousarray(instances, dtype=np.uint32)
        st = np.zeros(n, dtype=np.uint8); rv = np.zeros((n, 4), dtype=np.uint8)
        _lib.check(self._L.zkv_risc0_set_verify_batch(self._h, n, idx.ctypes.data, blob, off.ctypes.data, _cat32(image_ids, 'image_id'),
                                                      _cat32(journal_digests, 'journal_digest'), st.ctypes.data, rv.ctypes.data),
                   'zkv_risc0_set_verify_batch')
        return st, rv

    def verify_batch_dev(self, n, d_instances, d_seals, d_image_ids, d_journal_digests, d_status, d_recv=0, stream=0):
        _lib.check(self._L.zkv_risc0_set_verify_batch_dev(self._h, n, d_instances, d_seals, d_image_ids, d_journal_digests, d_status,
                                                          d_recv or None, stream or None), 'zkv_risc0_set_verify_batch_dev')

    def vk_x_batch(self, instances, var_signals):
        n = len(var_signals)
        _same_len(n, instances=instances)
        idx = np.ascontiguousarray(instances, dtype=np.uint32)
        blob = b''.join(bytes(a) + bytes(b) for a, b in var_signals) + b'\0'
        out = np.zeros(max(64 * n, 1), dtype=np.uint8)
        _lib.check(self._L.zkv_risc0_set_vk_x_batch(self._h, n, idx.ctypes.data, blob, out.ctypes.data), 'zkv_risc0_set_vk_x_batch')
        return [out[64 * i:64 * i + 64].tobytes() for i in range(n)]

    def set_aggregate_check(self, enable=True, seed=None, sub_batch=None):
        """Opt-in: share the pairing check among sub-batches of 16 ... 256 proofs (None: chosen by the failure rate seen) of a large chunk (include/zkv.h, csrc/zkv_agg.h);
        statuses stay the deterministic ones (a failed sub-batch is verified again proof by proof)."""
        _set_aggregate_check(self._L, self._h, enable, seed, sub_batch)

    def aggregate_counters(self):
        """(sub-batches checked in aggregate, sub-batches that failed and were verified proof by proof)."""
        return _aggregate_counters(self._L, self._h)

    def synchronize(self):
        _lib.check(self._L.zkv_ctx_synchronize(self._h), 'zkv_ctx_synchronize')
