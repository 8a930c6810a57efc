"""Per-proof status codes and the reference's error classes.

Reference: /root/reference/contracts/src/common/errors.rs:3-27, risc0/errors.rs:8-44, sp1/errors.rs:8-43.
The reference returns `Err(Vec<u8>)` holding the ABI-encoded Solidity custom error; `VerifierError.revert`
carries exactly those bytes (produced by zkv_status_abi_encode)."""
import ctypes as C

from . import _lib

STATUS_OK = 0
STATUS_VERIFICATION_FAILED = 1
STATUS_INVALID_INITIALIZATION = 2
STATUS_ALREADY_INITIALIZED = 3
STATUS_INVALID_PROOF_DATA = 4
STATUS_SELECTOR_MISMATCH = 5
STATUS_BAD_CALLDATA = 6          # wire layer only (wire.py): calldata the contract's router cannot decode

STATUS_NAMES = {
    0: 'Ok', 1: 'VerificationFailed', 2: 'InvalidInitialization', 3: 'AlreadyInitialized', 4: 'InvalidProofData',
    5: 'SelectorMismatch', 6: 'BadCalldata',
}
VM_RISC0, VM_SP1 = 0, 1


def revert_bytes(vm, status, received=b'\0\0\0\0', expected=b'\0\0\0\0'):
    out = C.create_string_buffer(68)
    n = _lib.lib().zkv_status_abi_encode(vm, status, bytes(received), bytes(expected), out)
    if n < 0:
        raise _lib.ZkvRuntimeError(n, 'zkv_status_abi_encode')
    return out.raw[:n]


class VerifierError(Exception):
    """`Err(Vec<u8>)` of the reference traits: `.status` is the error class, `.revert` the ABI-encoded bytes."""

    def __init__(self, vm, status, received=None, expected=None):
        self.vm, self.status, self.received, self.expected = vm, status, received, expected
        self.revert = revert_bytes(vm, status, received or b'\0' * 4, expected or b'\0' * 4)
        name = STATUS_NAMES.get(status, str(status))
        if status == STATUS_SELECTOR_MISMATCH:
            name = ('SelectorMismatch' if vm == VM_RISC0 else 'WrongVerifierSelector') + \
                '(received=%s, expected=%s)' % ((received or b'').hex(), (expected or b'').hex())
        super().__init__(name)
