"""Sharded (multi-device) verifiers: several single-device verifiers of one kind behind the same methods (include/zkv.h, "sharded
contexts"; SURVEY 8b `device_mask`).  Proofs are independent units -- the reference verifies one per call
(/root/reference/contracts/src/risc0/verifier.rs:78-92, sp1/verifier.rs:39-46) -- so a batch splits into contiguous ranges."""
import ctypes as C

from . import _lib


def shard(verifiers):
    """Wrap single-device verifiers (one per GPU; a GPU may carry several) into ONE verifier of the same class whose batch methods
    split every batch over them.  Takes ownership: the given objects are emptied.  All must be the same kind with the same parameters."""
    verifiers = list(verifiers)
    if not verifiers or any(type(v) is not type(verifiers[0]) for v in verifiers) or any(not getattr(v, '_h', None) for v in verifiers):
        raise ValueError('shard() needs live verifiers of one class')
    L = _lib.lib()
    arr = (C.c_void_p * len(verifiers))(*[v._h for v in verifiers])
    h = L.zkv_ctx_create_sharded(arr, len(verifiers))
    if not h:
        raise ValueError('zkv_ctx_create_sharded refused the shards (different kinds / parameters, or not initialised)')
    out = object.__new__(type(verifiers[0]))
    out.__dict__.update(verifiers[0].__dict__)
    out._h = h
    for v in verifiers:
        v._h = None
    return out


def shard_count(verifier):
    return int(_lib.lib().zkv_ctx_shard_count(verifier._h))


def shard_devices(verifier):
    L = _lib.lib()
    return [int(L.zkv_ctx_shard_device(verifier._h, k)) for k in range(shard_count(verifier))]


def shard_peer_access(verifier):
    """Per shard: 1 = peer access to the GPU that held the last staged batch's rows was granted (direct xGMI copies), 0 = refused (bounced
    copies), 2 = not applicable (same GPU, or nothing staged yet).  zkv_ctx_shard_peer_access."""
    L = _lib.lib()
    return [int(L.zkv_ctx_shard_peer_access(verifier._h, k)) for k in range(shard_count(verifier))]
