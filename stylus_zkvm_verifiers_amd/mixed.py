"""Mixed batches: one `VMType` tag per proof (/root/reference/contracts/src/common/types.rs:24-26; the reference's shared
Groth16 core takes the tag per call, common/groth16.rs:23-31, 96-103).

`MixedVerifier` is one `RiscZeroVerifier` (initialised with the given parameters) and one `Sp1Verifier` behind the tag:
proof i with vm[i] = VM_RISC0 is `IRiscZeroVerifier::verify(seal, image_id, journal_digest)`, with VM_SP1 it is
`ISp1Verifier::verify_proof(program_vkey, public_values, proof_bytes)`; statuses come back in the caller's order.  The
batch is demultiplexed on the device (include/zkv.h, "mixed batches")."""
import numpy as np

from . import _lib
from .errors import VM_RISC0, VM_SP1
from .risc0 import _aggregate_counters, _blob, _cat32, _same_len, _set_aggregate_check

STATUS_UNKNOWN_VM = 7


class MixedVerifier:
    def __init__(self, control_root, bn254_control_id, device=0):
        self._L = _lib.lib()
        if len(control_root) != 32 or len(bn254_control_id) != 32:
            raise ValueError('control_root and bn254_control_id must be 32 bytes')
        self._h = self._L.zkv_mixed_ctx_create(bytes(control_root), bytes(bn254_control_id), device)
        if not self._h:
            raise MemoryError('zkv_mixed_ctx_create')

    def close(self):
        if getattr(self, '_h', None):
            self._L.zkv_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def verify_batch(self, vm, seals, in_a, in_b):
        """vm: n tags (VM_RISC0 / VM_SP1); seals: n byte strings; in_a: n x 32 bytes (image id | program vkey);
        in_b: n byte strings (32-byte journal digest | public values).  Returns (status uint8[n], received selectors uint8[n,4])."""
        n = len(seals)
        _same_len(n, vm=vm, in_a=in_a, in_b=in_b)
        tags = np.ascontiguousarray(vm, dtype=np.uint8)
        sblob, soff = _blob(seals)
        bblob, boff = _blob(in_b)
        st = np.zeros(n, dtype=np.uint8); rv = np.zeros((n, 4), dtype=np.uint8)
        _lib.check(self._L.zkv_mixed_verify_batch(self._h, n, tags.ctypes.data, sblob, soff.ctypes.data, _cat32(in_a, 'in_a'), bblob,
                                                  boff.ctypes.data, st.ctypes.data, rv.ctypes.data), 'zkv_mixed_verify_batch')
        return st, rv

    def verify_batch_dev(self, n, d_vm, d_seals, d_in_a, d_in_b, b_stride, pv_len, d_status, d_recv=0, stream=0):
        """Everything resident in HBM (device pointers as ints): tags, 260-byte seals, 32-byte in_a rows, b_stride-byte in_b rows."""
        _lib.check(self._L.zkv_mixed_verify_batch_dev(self._h, n, d_vm, d_seals, d_in_a, d_in_b, b_stride, pv_len, d_status,
                                                      d_recv or None, stream or None), 'zkv_mixed_verify_batch_dev')

    def set_lanes_per_proof(self, lanes):
        """Kernel mapping of both verifiers behind the tag (0 automatic, 2, 16, 64, 128: see RiscZeroVerifier.set_lanes_per_proof)."""
        _lib.check(self._L.zkv_ctx_set_lanes_per_proof(self._h, lanes), 'zkv_ctx_set_lanes_per_proof')

    def reserve(self, n):
        _lib.check(self._L.zkv_ctx_reserve(self._h, n), 'zkv_ctx_reserve')

    def set_aggregate_check(self, enable=True, seed=None, sub_batch=None):
        """Opt-in: share the pairing check among sub-batches of 16 ... 256 proofs (None: chosen by the failure rate seen) of a large chunk (include/zkv.h, csrc/zkv_agg.h);
        statuses stay the deterministic ones (a failed sub-batch is verified again proof by proof)."""
        _set_aggregate_check(self._L, self._h, enable, seed, sub_batch)

    def aggregate_counters(self):
        """(sub-batches checked in aggregate, sub-batches that failed and were verified proof by proof)."""
        return _aggregate_counters(self._L, self._h)

    def synchronize(self):
        """Waits for everything this verifier enqueued -- on the MIXED context itself: an all-SP1 batch never sets up the RISC Zero
        child, and waiting on that child alone would return at once."""
        _lib.check(self._L.zkv_ctx_synchronize(self._h), 'zkv_ctx_synchronize')

    def last_stage_ms(self):
        import ctypes as C
        out = (C.c_float * 5)()
        _lib.check(self._L.zkv_ctx_last_stage_ms(self._h, out), 'zkv_ctx_last_stage_ms')
        return list(out)


__all__ = ['MixedVerifier', 'VM_RISC0', 'VM_SP1', 'STATUS_UNKNOWN_VM']
