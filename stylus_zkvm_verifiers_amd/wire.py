"""On-chain wire layer: `eth_call` calldata for the Solidity view of the two verifier traits.

Reference: the methods the example shells export (/root/reference/examples/risc0-verifier/src/lib.rs,
/root/reference/examples/sp1-verifier/src/lib.rs) under the signatures their clients use
(examples/risc0-verifier/examples/interact.rs:31-43, examples/sp1-verifier/examples/interact.rs:11-19).  Stylus exports
`Vec<u8>` as `uint8[]`: every seal byte is one 32-byte word.  Decoding runs on the device (csrc/k_wire.hip)."""
import ctypes as C

import numpy as np

from . import _lib
from .risc0 import _blob

RETURNDATA_STRIDE = 96
STATUS_BAD_CALLDATA = 6


def function_selector(signature):
    o = C.create_string_buffer(4)
    _lib.check(_lib.lib().zkv_abi_function_selector(signature.encode(), o), 'zkv_abi_function_selector')
    return o.raw


def _encode(fn, *args):
    need = fn(*args, None, 0)
    out = C.create_string_buffer(need)
    fn(*args, out, need)
    return out.raw


def encode_risc0_verify(seal, image_id, journal_digest):
    """calldata of `verify(uint8[] seal, bytes32 image_id, bytes32 journal_digest)` (interact.rs:36)"""
    return _encode(_lib.lib().zkv_risc0_encode_verify_call, bytes(seal), len(seal), bytes(image_id), bytes(journal_digest))


def encode_risc0_verify_integrity(seal, claim_digest):
    """calldata of `verifyIntegrity(uint8[] receipt_seal, bytes32 receipt_claim_digest)`"""
    return _encode(_lib.lib().zkv_risc0_encode_verify_integrity_call, bytes(seal), len(seal), bytes(claim_digest))


def encode_sp1_verify_proof(program_vkey, public_values, proof_bytes):
    """calldata of `verifyProof(bytes32 programVKey, uint8[] publicValues, uint8[] proofBytes)` (sp1 interact.rs:15)"""
    return _encode(_lib.lib().zkv_sp1_encode_verify_proof_call, bytes(program_vkey), bytes(public_values), len(public_values),
                   bytes(proof_bytes), len(proof_bytes))


def eth_call_batch(verifier, calldatas):
    """n eth_calls against `verifier` (RiscZeroVerifier or Sp1Verifier).  Returns (reverted uint8[n], returndata list of
    bytes, status uint8[n]); status is errors.STATUS_* for verify-class calls and STATUS_BAD_CALLDATA for the rest."""
    from .sp1 import Sp1Verifier
    L = _lib.lib()
    n = len(calldatas)
    blob, off = _blob(calldatas)
    rev = np.zeros(n, dtype=np.uint8); st = np.zeros(n, dtype=np.uint8)
    ret = np.zeros((max(n, 1), RETURNDATA_STRIDE), dtype=np.uint8); rl = np.zeros(max(n, 1), dtype=np.uint32)
    fn = L.zkv_sp1_eth_call_batch if isinstance(verifier, Sp1Verifier) else L.zkv_risc0_eth_call_batch
    _lib.check(fn(verifier._h, n, blob, off.ctypes.data, rev.ctypes.data, ret.ctypes.data, rl.ctypes.data, st.ctypes.data),
               'zkv_eth_call_batch')
    return rev, [ret[i, :rl[i]].tobytes() for i in range(n)], st


def eth_call_batch_dev(verifier, n, d_calldata, d_calldata_off, calldata_bytes, d_status, d_recv=0, stream=0):
    """Device-resident calldata blob + offsets (device pointers as ints); verify-class calls only; asynchronous."""
    _lib.check(_lib.lib().zkv_eth_call_batch_dev(verifier._h, n, d_calldata, d_calldata_off, calldata_bytes, d_status,
                                                 d_recv or None, stream or None), 'zkv_eth_call_batch_dev')


def last_wire_ms(verifier):
    out = C.c_float(0)
    _lib.check(_lib.lib().zkv_ctx_last_wire_ms(verifier._h, C.byref(out)), 'zkv_ctx_last_wire_ms')
    return out.value
