"""ctypes binding of the C oracle (oracle/zkv_oracle.c).  TEST INFRASTRUCTURE ONLY -- see the header of that file.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libzkv_oracle.so')


def build(force=False):
    src = os.path.join(HERE, 'zkv_oracle.c')
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', HERE, 'libzkv_oracle.so'], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        u8p = C.POINTER(C.c_uint8)
        L.zkvo_risc0_new.restype = C.c_void_p
        L.zkvo_risc0_free.argtypes = [C.c_void_p]
        L.zkvo_risc0_initialize.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.zkvo_risc0_get_selector.argtypes = [C.c_void_p, C.c_char_p]
        L.zkvo_risc0_get_control_root.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.zkvo_risc0_is_initialized.argtypes = [C.c_void_p]
        L.zkvo_risc0_verify.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p]
        L.zkvo_risc0_verify_integrity.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p]
        L.zkvo_sp1_verify_proof.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p]
        L.zkvo_ecadd.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.zkvo_ecmul.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.zkvo_ecpairing.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.zkvo_g2_classify.argtypes = [C.c_char_p]
        L.zkvo_sha256.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.zkvo_risc0_vk_digest.argtypes = [C.c_char_p]
        L.zkvo_risc0_claim_digest.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        L.zkvo_sp1_hash_public_values.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.zkvo_sp1_verifier_hash.argtypes = [C.c_char_p]
        L.zkvo_sp1_version.restype = C.c_char_p
        L.zkvo_groth16_vk_x.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_char_p]
        L.zkvo_groth16_verify_vk.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int]
        L.zkvo_groth16_vk_x_vk.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p]
        L.zkvo_plonk_verify.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_int]
        L.zkvo_sp1_plonk_verify_proof.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p]
        L.zkvo_sp1_plonk_verify_batch.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_int]
        L.zkvo_status_abi_encode.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
        L.zkvo_count_enable.argtypes = [C.c_int]
        L.zkvo_count_read.restype = C.c_uint64
        L.zkvo_fp_mulmod.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        L.zkvo_risc0_verify_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_int]
        L.zkvo_sp1_verify_batch.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_int]
        L.zkvo_keccak256.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.zkvo_risc0_encode_call.restype = C.c_size_t
        L.zkvo_risc0_encode_call.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p]
        L.zkvo_sp1_encode_call.restype = C.c_size_t
        L.zkvo_sp1_encode_call.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p]
        L.zkvo_risc0_eth_call.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.zkvo_sp1_eth_call.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _buf(n):
    return C.create_string_buffer(n)


class Risc0Oracle:
    """Mirror of IRiscZeroVerifier (risc0/verifier.rs:18-42) over the C oracle."""

    def __init__(self):
        self._h = lib().zkvo_risc0_new()

    def __del__(self):
        try:
            lib().zkvo_risc0_free(self._h)
        except Exception:
            pass

    def initialize(self, control_root, bn254_control_id):
        return lib().zkvo_risc0_initialize(self._h, bytes(control_root), bytes(bn254_control_id))

    def verify(self, seal, image_id, journal_digest):
        r = _buf(4)
        st = lib().zkvo_risc0_verify(self._h, bytes(seal), len(seal), bytes(image_id), bytes(journal_digest), r)
        return st, (r.raw if st == 5 else None)

    def verify_integrity(self, seal, claim_digest):
        r = _buf(4)
        st = lib().zkvo_risc0_verify_integrity(self._h, bytes(seal), len(seal), bytes(claim_digest), r)
        return st, (r.raw if st == 5 else None)

    def get_selector(self):
        r = _buf(4); lib().zkvo_risc0_get_selector(self._h, r); return r.raw

    def get_control_root(self):
        a, b = _buf(16), _buf(16); lib().zkvo_risc0_get_control_root(self._h, a, b); return a.raw, b.raw

    def is_initialized(self):
        return bool(lib().zkvo_risc0_is_initialized(self._h))

    def eth_call(self, calldata):
        """One eth_call against the RISC Zero shell: (reverted, returndata, status or None)."""
        ret = _buf(96); n = C.c_size_t(0); st = C.c_int(0)
        rev = lib().zkvo_risc0_eth_call(self._h, bytes(calldata), len(calldata), ret, C.byref(n), C.byref(st))
        return bool(rev), ret.raw[:n.value], (None if st.value < 0 else st.value)

    def verify_batch(self, seals, image_ids, journal_digests, threads=1):
        """seals: list of bytes; image_ids/journal_digests: list of 32-byte values. Returns (status bytes, recv bytes)."""
        import numpy as np
        n = len(seals)
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(s) for s in seals])
        blob = b''.join(seals) + b'\0'
        ids = b''.join(image_ids); jds = b''.join(journal_digests)
        st = np.zeros(n, dtype=np.uint8); rv = np.zeros(4 * n, dtype=np.uint8)
        lib().zkvo_risc0_verify_batch(self._h, n, blob, off.ctypes.data, ids, jds, st.ctypes.data, rv.ctypes.data, threads)
        return st, rv


def sp1_verify_proof(vkey, public_values, proof):
    r = _buf(4)
    st = lib().zkvo_sp1_verify_proof(bytes(vkey), bytes(public_values), len(public_values), bytes(proof), len(proof), r)
    return st, (r.raw if st == 5 else None)


def sp1_verify_batch(vkeys, pvs, proofs, threads=1):
    import numpy as np
    n = len(proofs)
    poff = np.zeros(n + 1, dtype=np.uint64); poff[1:] = np.cumsum([len(s) for s in proofs])
    voff = np.zeros(n + 1, dtype=np.uint64); voff[1:] = np.cumsum([len(s) for s in pvs])
    st = np.zeros(n, dtype=np.uint8); rv = np.zeros(4 * n, dtype=np.uint8)
    lib().zkvo_sp1_verify_batch(n, b''.join(vkeys), b''.join(pvs) + b'\0', voff.ctypes.data, b''.join(proofs) + b'\0',
                                poff.ctypes.data, st.ctypes.data, rv.ctypes.data, threads)
    return st, rv


def ecadd(data):
    o = _buf(64); return o.raw if lib().zkvo_ecadd(bytes(data), len(data), o) else None


def ecmul(data):
    o = _buf(64); return o.raw if lib().zkvo_ecmul(bytes(data), len(data), o) else None


def ecpairing(data):
    o = _buf(32); return o.raw if lib().zkvo_ecpairing(bytes(data), len(data), o) else None


def g2_classify(data128):
    return lib().zkvo_g2_classify(bytes(data128))


def sha256(b):
    o = _buf(32); lib().zkvo_sha256(bytes(b), len(b), o); return o.raw


def risc0_vk_digest():
    o = _buf(32); lib().zkvo_risc0_vk_digest(o); return o.raw


def risc0_claim_digest(image_id, journal_digest):
    o = _buf(32); lib().zkvo_risc0_claim_digest(bytes(image_id), bytes(journal_digest), o); return o.raw


def sp1_hash_public_values(pv):
    o = _buf(32); lib().zkvo_sp1_hash_public_values(bytes(pv), len(pv), o); return o.raw


def groth16_vk_x(vm, signals):
    o = _buf(64)
    ok = lib().zkvo_groth16_vk_x(vm, b''.join(signals), len(signals), o)
    return o.raw if ok else None


def groth16_vk_x_vk(vk_words, n_ic, signals):
    """compute_vk_x (groth16.rs:51-58) for an arbitrary key: 64-byte affine point, None when a precompile call fails."""
    o = _buf(64)
    ok = lib().zkvo_groth16_vk_x_vk(bytes(vk_words), n_ic, b''.join(signals) + b'\0', len(signals), o)
    return o.raw if ok else None


def groth16_verify_vk(vm, vk_words, n_ic, proof_words, signals):
    """verify_proof_with_key for an arbitrary key: vk_words bytes, proof_words 256 bytes, signals list of 32-byte values."""
    return bool(lib().zkvo_groth16_verify_vk(vm, bytes(vk_words), n_ic, bytes(proof_words), b''.join(signals) + b'\0', len(signals)))


def plonk_verify(vk_bytes, proof_words, public_inputs):
    """gnark-style BN254 PLONK verification (zkv_plonk_oracle.inc): proof 27 x 32 bytes, public inputs as 32-byte values -> bool."""
    return bool(lib().zkvo_plonk_verify(bytes(vk_bytes), len(vk_bytes), bytes(proof_words), len(proof_words), b''.join(public_inputs) + b'\0', len(public_inputs)))


def sp1_plonk_verify_proof(vk_bytes, verifier_hash, vkey, public_values, proof):
    recv = _buf(4)
    st = lib().zkvo_sp1_plonk_verify_proof(bytes(vk_bytes), len(vk_bytes), bytes(verifier_hash), bytes(vkey), bytes(public_values) + b'\0', len(public_values),
                                           bytes(proof) + b'\0', len(proof), recv)
    return st, recv.raw


def sp1_plonk_verify_batch(vk_bytes, verifier_hash, vkeys, public_values, proofs, threads=1):
    import numpy as np
    n = len(proofs)
    pvb, pvo = _blob_np(public_values)
    pb, po = _blob_np(proofs)
    st = np.zeros(n, dtype=np.uint8); rv = np.zeros(4 * max(n, 1), dtype=np.uint8)
    lib().zkvo_sp1_plonk_verify_batch(bytes(vk_bytes), len(vk_bytes), bytes(verifier_hash), n, b''.join(bytes(v) for v in vkeys) + b'\0', pvb, pvo.ctypes.data, pb,
                                      po.ctypes.data, st.ctypes.data, rv.ctypes.data, threads)
    return st, rv[:4 * n]


def _blob_np(items):
    import numpy as np
    off = np.zeros(len(items) + 1, dtype=np.uint64)
    if len(items):
        off[1:] = np.cumsum([len(s) for s in items], dtype=np.uint64)
    return b''.join(bytes(s) for s in items) + b'\0', off


def status_abi_encode(vm, status, recv, exp):
    o = _buf(68)
    n = lib().zkvo_status_abi_encode(vm, status, bytes(recv), bytes(exp), o)
    return o.raw[:n] if n >= 0 else None


def fp_mulmod(a, b):
    o = _buf(32); lib().zkvo_fp_mulmod(bytes(a), bytes(b), o); return o.raw


def keccak256(b):
    o = _buf(32); lib().zkvo_keccak256(bytes(b), len(b), o); return o.raw


def risc0_encode_call(seal, a, b=None):
    """Canonical calldata of verify(seal, a, b) or, with b None, verifyIntegrity(seal, a)."""
    n = lib().zkvo_risc0_encode_call(int(b is None), bytes(seal), len(seal), bytes(a), bytes(b or bytes(32)), None)
    o = _buf(n); lib().zkvo_risc0_encode_call(int(b is None), bytes(seal), len(seal), bytes(a), bytes(b or bytes(32)), o)
    return o.raw


def sp1_encode_call(vkey, pv, proof):
    n = lib().zkvo_sp1_encode_call(bytes(vkey), bytes(pv), len(pv), bytes(proof), len(proof), None)
    o = _buf(n); lib().zkvo_sp1_encode_call(bytes(vkey), bytes(pv), len(pv), bytes(proof), len(proof), o)
    return o.raw


def sp1_eth_call(calldata):
    ret = _buf(96); n = C.c_size_t(0); st = C.c_int(0)
    rev = lib().zkvo_sp1_eth_call(bytes(calldata), len(calldata), ret, C.byref(n), C.byref(st))
    return bool(rev), ret.raw[:n.value], (None if st.value < 0 else st.value)
