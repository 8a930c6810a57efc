"""Host-side mirror of `Groth16Verifier::verify_proof_with_key` (/root/reference/contracts/src/common/groth16.rs:23-49) for an
arbitrary verification key (`VerificationKey`, common/types.rs:17-23)."""
import numpy as np

from . import _lib
from .errors import VM_RISC0, VM_SP1
from .risc0 import _aggregate_counters, _set_aggregate_check


def vk_words(alpha1, beta2, gamma2, delta2, ic):
    """Serialise a key given as integers in the reference's layout: alpha1 (x, y); beta2/gamma2/delta2 ((x0, x1), (y0, y1)) with
    index 0 = imaginary, 1 = real; ic list of (x, y)."""
    be = lambda v: int(v).to_bytes(32, 'big')
    out = be(alpha1[0]) + be(alpha1[1])
    for q in (beta2, gamma2, delta2):
        out += be(q[0][0]) + be(q[0][1]) + be(q[1][0]) + be(q[1][1])
    for x, y in ic:
        out += be(x) + be(y)
    return out


class Groth16Verifier:
    def __init__(self, vk_bytes, n_ic, vm_type=VM_SP1, device=0):
        if len(vk_bytes) != 448 + 64 * n_ic:
            raise ValueError('verification key must be 448 + 64 * n_ic bytes')
        self._L = _lib.lib()
        self.n_ic = n_ic
        self._h = self._L.zkv_groth16_ctx_create(bytes(vk_bytes), n_ic, vm_type, device)
        if not self._h:
            raise ValueError('zkv_groth16_ctx_create rejected the arguments')

    def close(self):
        if getattr(self, '_h', None):
            self._L.zkv_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def verify_batch(self, proofs, signals):
        """proofs: list of 256-byte (a, b, c) word blocks; signals: list of lists of 32-byte big-endian values -> bool array."""
        n = len(proofs)
        if len(signals) != n:
            raise ValueError('signals has %d entries for a batch of %d proofs' % (len(signals), n))
        for p in proofs:
            if len(p) != 256:
                raise ValueError('a proof is 8 x 32 bytes')
        for s in signals:
            if len(s) != self.n_ic - 1:
                raise ValueError('expected %d signals per proof' % (self.n_ic - 1))   # groth16.rs:32 length check
        pb = b''.join(bytes(p) for p in proofs) + b'\0'
        sb = b''.join(b''.join(bytes(x) for x in s) for s in signals) + b'\0'
        out = np.zeros(max(n, 1), dtype=np.uint8)
        _lib.check(self._L.zkv_groth16_verify_batch(self._h, n, pb, sb, out.ctypes.data), 'zkv_groth16_verify_batch')
        return out[:n].astype(bool)

    def set_aggregate_check(self, enable=True, seed=None, sub_batch=None):
        """Opt-in: share the pairing check among sub-batches of a large chunk (include/zkv.h); the answers stay the deterministic ones."""
        _set_aggregate_check(self._L, self._h, enable, seed, sub_batch)

    def aggregate_counters(self):
        return _aggregate_counters(self._L, self._h)

    def vk_x_batch(self, signals):
        """Groth16Verifier::compute_vk_x (common/groth16.rs:51-58): signals = list of n_ic - 1 32-byte big-endian values per proof
        (each < R); returns the 64-byte affine vk_x per proof ((0,0) = infinity)."""
        n = len(signals)
        for s in signals:
            if len(s) != self.n_ic - 1:
                raise ValueError('expected %d signals per proof' % (self.n_ic - 1))
        sb = b''.join(b''.join(bytes(x) for x in s) for s in signals) + b'\0'
        out = np.zeros(max(64 * n, 1), dtype=np.uint8)
        _lib.check(self._L.zkv_ctx_vk_x_batch(self._h, n, sb, out.ctypes.data), 'zkv_ctx_vk_x_batch')
        return [out[64 * i:64 * i + 64].tobytes() for i in range(n)]

    def verify_proof_with_key(self, a, b, c, public_signals):
        """Single call with the reference's argument shapes: a [x, y], b [[x0, x1], [y0, y1]], c [x, y], signals as ints."""
        be = lambda v: int(v).to_bytes(32, 'big')
        proof = b''.join(be(v) for v in (a[0], a[1], b[0][0], b[0][1], b[1][0], b[1][1], c[0], c[1]))
        if len(public_signals) + 1 != self.n_ic or any(int(s) >= (1 << 256) for s in public_signals):
            return False
        return bool(self.verify_batch([proof], [[be(s) for s in public_signals]])[0])
