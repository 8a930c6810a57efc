"""Precompile-level batches: the EVM precompiles the reference calls (common/groth16.rs:12-14, 54-55, 121-125) with
EIP-196/197 semantics, many calls per launch.  A failed precompile call (the reference's Err(())) is reported as None."""
import numpy as np

from . import _lib


class Bn254Precompiles:
    def __init__(self, device=0):
        self._L = _lib.lib()
        self._h = self._L.zkv_bn254_ctx_create(device)
        if not self._h:
            raise MemoryError('zkv_bn254_ctx_create')

    def close(self):
        if getattr(self, '_h', None):
            self._L.zkv_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def _run(self, fn, name, inputs, in_sz, out_sz, *extra):
        n = len(inputs)
        for x in inputs:
            if len(x) != in_sz:
                raise ValueError('%s input must be %d bytes' % (name, in_sz))
        blob = b''.join(bytes(x) for x in inputs) + b'\0'
        out = np.zeros(max(n * out_sz, 1), dtype=np.uint8); ok = np.zeros(max(n, 1), dtype=np.uint8)
        _lib.check(fn(self._h, n, *extra, blob, out.ctypes.data, ok.ctypes.data), name)
        return out, ok

    def ecadd(self, inputs):
        """inputs: 128-byte calls -> list of 64-byte results (None where the precompile fails)."""
        out, ok = self._run(self._L.zkv_bn254_ecadd_batch, 'zkv_bn254_ecadd_batch', inputs, 128, 64)
        return [out[64 * i:64 * i + 64].tobytes() if ok[i] else None for i in range(len(inputs))]

    def ecmul(self, inputs):
        out, ok = self._run(self._L.zkv_bn254_ecmul_batch, 'zkv_bn254_ecmul_batch', inputs, 96, 64)
        return [out[64 * i:64 * i + 64].tobytes() if ok[i] else None for i in range(len(inputs))]

    def pairing_dev(self, n, k, d_in, d_result, d_ok, stream=0):
        """n calls of k pairs resident in HBM (device pointers as ints): result / ok bytes are written by the kernels on `stream`."""
        _lib.check(self._L.zkv_bn254_pairing_batch_dev(self._h, n, k, d_in, d_result, d_ok, stream or None), 'zkv_bn254_pairing_batch_dev')

    def synchronize(self):
        _lib.check(self._L.zkv_ctx_synchronize(self._h), 'zkv_ctx_synchronize')

    def pairing(self, inputs, k):
        """inputs: calls of k pairs (k*192 bytes each) -> list of True/False (None where the precompile fails)."""
        out, ok = self._run(self._L.zkv_bn254_pairing_batch, 'zkv_bn254_pairing_batch', inputs, 192 * k, 1, k)
        return [bool(out[i]) if ok[i] else None for i in range(len(inputs))]
