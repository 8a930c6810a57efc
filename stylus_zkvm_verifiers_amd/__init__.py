"""MI355X-native batch Groth16 (BN254) verification for RISC Zero v2.1 and SP1 v5.0.0 proofs.

Drop-in for the verify path of gnosisguild/stylus-zkvm-verifiers: `RiscZeroVerifier` mirrors
`IRiscZeroVerifier`, `Sp1Verifier` mirrors `ISp1Verifier`; all arithmetic runs in hand-written HIP kernels
(libzkv_mi355x.so, C ABI in include/zkv.h).  There is no CPU fallback."""
from . import errors
from .errors import VerifierError
from .risc0 import RiscZeroVerifier, RiscZeroVerifierSet
from .sp1 import Sp1Verifier, Sp1PlonkVerifier
from .bn254 import Bn254Precompiles
from .groth16 import Groth16Verifier
from .mixed import MixedVerifier
from . import wire
from .sharded import shard, shard_count, shard_devices, shard_peer_access

__all__ = ['host_register', 'host_unregister', 'RiscZeroVerifier', 'RiscZeroVerifierSet', 'Sp1Verifier', 'Sp1PlonkVerifier', 'Bn254Precompiles', 'Groth16Verifier', 'MixedVerifier', 'VerifierError', 'errors', 'wire', 'device_count', 'shard', 'shard_count', 'shard_devices', 'shard_peer_access']


def device_count():
    from . import _lib
    return _lib.lib().zkv_device_count()


def _addr_bytes(buf):
    import numpy as np
    a = buf if isinstance(buf, np.ndarray) else np.frombuffer(buf, dtype=np.uint8)
    if not a.flags['C_CONTIGUOUS']:
        raise ValueError('host_register needs a contiguous buffer')
    return a.ctypes.data, a.nbytes


def host_register(buf):
    """Pins a host buffer (numpy array / writable bytes-like) that is handed to the batch entry points repeatedly (zkv_host_register):
    the H2D staging of host-buffer batches then is direct DMA.  The buffer must stay alive until host_unregister(buf)."""
    from . import _lib
    p, n = _addr_bytes(buf)
    _lib.check(_lib.lib().zkv_host_register(p, n), 'zkv_host_register')


def host_unregister(buf):
    from . import _lib
    p, _ = _addr_bytes(buf)
    _lib.check(_lib.lib().zkv_host_unregister(p), 'zkv_host_unregister')
