"""Host-side mirror of `ISp1Verifier` / `Sp1Verifier`
(/root/reference/contracts/src/sp1/verifier.rs:16-111) over the C ABI of libzkv_mi355x.so."""
import ctypes as C

import numpy as np

from . import _lib
from .errors import STATUS_OK, STATUS_SELECTOR_MISMATCH, VM_SP1, VerifierError
from .risc0 import _aggregate_counters, _blob, _cat32, _same_len, _set_aggregate_check


class Sp1Verifier:
    def __init__(self, device=0):
        self._L = _lib.lib()
        self._h = self._L.zkv_sp1_ctx_create(device)
        if not self._h:
            raise MemoryError('zkv_sp1_ctx_create')

    def close(self):
        if getattr(self, '_h', None):
            self._L.zkv_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- ISp1Verifier
    def verify_proof(self, program_vkey, public_values, proof_bytes):
        """sp1/verifier.rs:39-46, 58-111: returns None or raises VerifierError."""
        st = C.c_uint8(0); rv = C.create_string_buffer(4)
        _lib.check(self._L.zkv_sp1_verify_proof(self._h, bytes(program_vkey), bytes(public_values), len(public_values),
                                                bytes(proof_bytes), len(proof_bytes), C.byref(st), rv), 'zkv_sp1_verify_proof')
        if st.value == STATUS_OK:
            return None
        if st.value == STATUS_SELECTOR_MISMATCH:
            raise VerifierError(VM_SP1, st.value, rv.raw, self.verifier_hash()[:4])
        raise VerifierError(VM_SP1, st.value)

    def verifier_hash(self):
        o = C.create_string_buffer(32); self._L.zkv_sp1_verifier_hash(o); return o.raw

    def version(self):
        return self._L.zkv_sp1_version().decode()

    # ---- batch
    def verify_batch(self, program_vkeys, public_values, proofs):
        n = len(proofs)
        _same_len(n, program_vkeys=program_vkeys, public_values=public_values)
        pblob, poff = _blob(proofs)
        vblob, voff = _blob(public_values)
        st = np.zeros(n, dtype=np.uint8); rv = np.zeros((n, 4), dtype=np.uint8)
        _lib.check(self._L.zkv_sp1_verify_batch(self._h, n, _cat32(program_vkeys, 'program_vkey'), vblob, voff.ctypes.data,
                                                pblob, poff.ctypes.data, st.ctypes.data, rv.ctypes.data), 'zkv_sp1_verify_batch')
        return st, rv

    def verify_batch_dev(self, n, d_vkeys, d_public_values, pv_len, d_proofs, d_status, d_recv=0, stream=0):
        _lib.check(self._L.zkv_sp1_verify_batch_dev(self._h, n, d_vkeys, d_public_values, pv_len, d_proofs, d_status,
                                                    d_recv or None, stream or None), 'zkv_sp1_verify_batch_dev')

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


class Sp1PlonkVerifier:
    """`ISp1Verifier` (/root/reference/contracts/src/sp1/verifier.rs:16-29) with the PLONK proof system behind `verify_proof` -- the
    path the reference marks "in progress" (README.md:25) and holds no code for: PARITY UNPINNED BY CONSTRUCTION.  The verifying key
    (include/zkv.h, "SP1 PLONK verifier") and the 32-byte verifier hash are supplied by the caller."""

    def __init__(self, vk_bytes, verifier_hash, device=0):
        self._L = _lib.lib()
        if len(verifier_hash) != 32:
            raise ValueError('verifier_hash must be 32 bytes')
        self._hash = bytes(verifier_hash)
        self._h = self._L.zkv_sp1_plonk_ctx_create(bytes(vk_bytes), len(vk_bytes), self._hash, device)
        if not self._h:
            raise ValueError('zkv_sp1_plonk_ctx_create rejected the verifying key')

    def close(self):
        if getattr(self, '_h', None):
            self._L.zkv_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def verifier_hash(self):
        return self._hash

    def verify_proof(self, program_vkey, public_values, proof_bytes):
        """Returns None or raises VerifierError, like Sp1Verifier.verify_proof."""
        st = C.c_uint8(0); rv = C.create_string_buffer(4)
        _lib.check(self._L.zkv_sp1_plonk_verify_proof(self._h, bytes(program_vkey), bytes(public_values), len(public_values),
                                                      bytes(proof_bytes), len(proof_bytes), C.byref(st), rv), 'zkv_sp1_plonk_verify_proof')
        if st.value == STATUS_OK:
            return None
        if st.value == STATUS_SELECTOR_MISMATCH:
            raise VerifierError(VM_SP1, st.value, rv.raw, self._hash[:4])
        raise VerifierError(VM_SP1, st.value)

    def verify_batch(self, program_vkeys, public_values, proofs):
        n = len(proofs)
        _same_len(n, program_vkeys=program_vkeys, public_values=public_values)
        pblob, poff = _blob(proofs)
        vblob, voff = _blob(public_values)
        st = np.zeros(n, dtype=np.uint8); rv = np.zeros((n, 4), dtype=np.uint8)
        _lib.check(self._L.zkv_sp1_plonk_verify_batch(self._h, n, _cat32(program_vkeys, 'program_vkey'), vblob, voff.ctypes.data,
                                                      pblob, poff.ctypes.data, st.ctypes.data, rv.ctypes.data), 'zkv_sp1_plonk_verify_batch')
        return st, rv

    def verify_batch_dev(self, n, d_vkeys, d_public_values, pv_len, d_proofs, d_status, d_recv=0, stream=0):
        _lib.check(self._L.zkv_sp1_plonk_verify_batch_dev(self._h, n, d_vkeys, d_public_values, pv_len, d_proofs, d_status,
                                                          d_recv or None, stream or None), 'zkv_sp1_plonk_verify_batch_dev')

    def set_lanes_per_proof(self, lanes):
        _lib.check(self._L.zkv_ctx_set_lanes_per_proof(self._h, lanes), 'zkv_ctx_set_lanes_per_proof')

    def reserve(self, n):
        _lib.check(self._L.zkv_ctx_reserve(self._h, n), 'zkv_ctx_reserve')

    def set_aggregate_check(self, enable=True, seed=None, sub_batch=None):
        """Opt-in (include/zkv.h): sub-batches share the two-pair check -- e(sum r D, [1]_2) e(sum r (-Q), [tau]_2) -- so no per-proof Miller
        loop or final exponentiation is left; statuses stay the per-proof ones (failed sub-batches are checked again proof by proof)."""
        _set_aggregate_check(self._L, self._h, enable, seed, sub_batch)

    def aggregate_counters(self):
        return _aggregate_counters(self._L, self._h)

    def synchronize(self):
        _lib.check(self._L.zkv_ctx_synchronize(self._h), 'zkv_ctx_synchronize')

    def last_stage_ms(self):
        out = (C.c_float * 5)()
        _lib.check(self._L.zkv_ctx_last_stage_ms(self._h, out), 'zkv_ctx_last_stage_ms')
        return list(out)
