"""ctypes binding of libzkv_mi355x.so (include/zkv.h).  Fails loudly when the HIP library is missing:
there is no CPU fallback in this package."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ZKV_LIB_PATH') or os.path.join(HERE, 'libzkv_mi355x.so')     # ZKV_LIB_PATH: A/B builds of the same library (tools/)

OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_WRONG_CTX = 0, -1, -2, -3, -4, -5
_ERR_NAMES = {ERR_INVALID_ARG: 'ZKV_ERR_INVALID_ARG', ERR_NO_DEVICE: 'ZKV_ERR_NO_DEVICE (no usable gfx950 device; there is no CPU fallback)',
              ERR_HIP: 'ZKV_ERR_HIP', ERR_OOM: 'ZKV_ERR_OOM', ERR_WRONG_CTX: 'ZKV_ERR_WRONG_CTX'}

# every symbol include/zkv.h declares: name -> (restype, argtypes)
_vp, _cp, _sz, _i, _u8p = C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint8)
SYMBOLS = {
    'zkv_device_count': (_i, []),
    'zkv_version': (_cp, []),
    'zkv_risc0_ctx_new': (_vp, [_i]),
    'zkv_risc0_initialize': (_i, [_vp, _cp, _cp, _u8p]),
    'zkv_risc0_ctx_create': (_vp, [_cp, _cp, _i]),
    'zkv_ctx_destroy': (None, [_vp]),
    'zkv_risc0_get_selector': (_i, [_vp, _cp]),
    'zkv_risc0_get_control_root': (_i, [_vp, _cp, _cp]),
    'zkv_risc0_get_bn254_control_id': (_i, [_vp, _cp]),
    'zkv_risc0_get_verifier_key_digest': (_i, [_vp, _cp]),
    'zkv_risc0_is_initialized': (_i, [_vp]),
    'zkv_risc0_verify_batch': (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    'zkv_risc0_verify_integrity_batch': (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    'zkv_risc0_verify': (_i, [_vp, _cp, _sz, _cp, _cp, _u8p, _cp]),
    'zkv_risc0_verify_integrity': (_i, [_vp, _cp, _sz, _cp, _u8p, _cp]),
    'zkv_risc0_verify_batch_dev': (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    'zkv_risc0_set_create': (_vp, [_sz, _cp, _cp, _i]),
    'zkv_risc0_set_size': (_sz, [_vp]),
    'zkv_risc0_set_get_selector': (_i, [_vp, _sz, _cp]),
    'zkv_risc0_set_verify_batch': (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'zkv_risc0_set_verify_batch_dev': (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'zkv_risc0_set_vk_x_batch': (_i, [_vp, _sz, _vp, _vp, _vp]),
    'zkv_sp1_ctx_create': (_vp, [_i]),
    'zkv_sp1_verifier_hash': (_i, [_cp]),
    'zkv_sp1_version': (_cp, []),
    'zkv_sp1_verify_batch': (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'zkv_sp1_verify_proof': (_i, [_vp, _cp, _cp, _sz, _cp, _sz, _u8p, _cp]),
    'zkv_sp1_verify_batch_dev': (_i, [_vp, _sz, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    'zkv_mixed_ctx_create': (_vp, [_cp, _cp, _i]),
    'zkv_mixed_ctx_risc0': (_vp, [_vp]),
    'zkv_mixed_ctx_sp1': (_vp, [_vp]),
    'zkv_mixed_verify_batch': (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'zkv_mixed_verify_batch_dev': (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    'zkv_sp1_plonk_ctx_create': (_vp, [_cp, _sz, _cp, _i]),
    'zkv_sp1_plonk_verifier_hash': (_i, [_vp, _cp]),
    'zkv_sp1_plonk_verify_batch': (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'zkv_sp1_plonk_verify_proof': (_i, [_vp, _cp, _cp, _sz, _cp, _sz, _u8p, _cp]),
    'zkv_sp1_plonk_verify_batch_dev': (_i, [_vp, _sz, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    'zkv_bn254_ctx_create': (_vp, [_i]),
    'zkv_bn254_ecadd_batch': (_i, [_vp, _sz, _vp, _vp, _vp]),
    'zkv_bn254_ecmul_batch': (_i, [_vp, _sz, _vp, _vp, _vp]),
    'zkv_bn254_pairing_batch': (_i, [_vp, _sz, _sz, _vp, _vp, _vp]),
    'zkv_bn254_pairing_batch_dev': (_i, [_vp, _sz, _sz, _vp, _vp, _vp, _vp]),
    'zkv_groth16_ctx_create': (_vp, [_cp, _sz, _i, _i]),
    'zkv_groth16_verify_batch': (_i, [_vp, _sz, _vp, _vp, _vp]),
    'zkv_ctx_vk_x_batch': (_i, [_vp, _sz, _vp, _vp]),
    'zkv_diag_mulmod_rate': (_i, [_i, _i, _i, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'zkv_diag_issue_rate': (_i, [_i, _i, _i, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'zkv_ctx_create_sharded': (_vp, [C.POINTER(_vp), _sz]),
    'zkv_ctx_shard_count': (_sz, [_vp]),
    'zkv_ctx_shard_device': (_i, [_vp, _sz]),
    'zkv_risc0_ctx_create_multi': (_vp, [_cp, _cp, C.c_uint64]),
    'zkv_sp1_ctx_create_multi': (_vp, [C.c_uint64]),
    'zkv_mixed_ctx_create_multi': (_vp, [_cp, _cp, C.c_uint64]),
    'zkv_ctx_vm': (_i, [_vp]),
    'zkv_ctx_set_lanes_per_proof': (_i, [_vp, _i]),
    'zkv_ctx_set_aggregate_check': (_i, [_vp, _i, _cp]),
    'zkv_ctx_aggregate_counters': (_i, [_vp, C.POINTER(C.c_uint64)]),
    'zkv_ctx_reserve': (_i, [_vp, _sz]),
    'zkv_diag_wait_faults': (_i, [_i, C.POINTER(C.c_uint64)]),
    'zkv_ctx_shard_peer_access': (_i, [_vp, _sz]),
    'zkv_host_register': (_i, [_vp, _sz]),
    'zkv_host_unregister': (_i, [_vp]),
    'zkv_chunk_capacity': (_sz, []),
    'zkv_ctx_synchronize': (_i, [_vp]),
    'zkv_ctx_last_stage_ms': (_i, [_vp, C.POINTER(C.c_float)]),
    'zkv_status_abi_encode': (_i, [_i, C.c_uint8, _cp, _cp, _cp]),
    'zkv_abi_function_selector': (_i, [_cp, _cp]),
    'zkv_risc0_encode_verify_call': (_sz, [_cp, _sz, _cp, _cp, _cp, _sz]),
    'zkv_risc0_encode_verify_integrity_call': (_sz, [_cp, _sz, _cp, _cp, _sz]),
    'zkv_sp1_encode_verify_proof_call': (_sz, [_cp, _cp, _sz, _cp, _sz, _cp, _sz]),
    'zkv_risc0_eth_call_batch': (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    'zkv_sp1_eth_call_batch': (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    'zkv_eth_call_batch_dev': (_i, [_vp, _sz, _vp, _vp, C.c_uint64, _vp, _vp, _vp]),
    'zkv_eth_call_returndata': (_i, [_vp, C.c_uint8, _cp, _cp, C.POINTER(C.c_uint32), _u8p]),
    'zkv_ctx_last_wire_ms': (_i, [_vp, C.POINTER(C.c_float)]),
}


class ZkvRuntimeError(RuntimeError):
    """A library/runtime failure (HIP, no device, bad arguments) -- never a verification outcome."""

    def __init__(self, code, where):
        super().__init__('%s failed: %s' % (where, _ERR_NAMES.get(code, 'error %d' % code)))
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError('libzkv_mi355x.so is not built (%s); run `python -m stylus_zkvm_verifiers_amd.build` -- '
                              'this package has no CPU fallback' % LIB_PATH)
        # PyTorch-ROCm bundles its own HIP runtime (another soname than /opt/rocm's, which this library links).  Both can live in one
        # process only when torch's is initialised first -- otherwise torch later reports "No HIP GPUs are available" -- so the
        # binding pulls torch in before the library whenever torch is installed (it is the package's plumbing for device memory).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)          # AttributeError when a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, where):
    if rc != OK:
        raise ZkvRuntimeError(rc, where)
