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
from .sharded import shard, shard_count, shard_devices

__all__ = ['RiscZeroVerifier', 'RiscZeroVerifierSet', 'Sp1Verifier', 'Sp1PlonkVerifier', 'Bn254Precompiles', 'Groth16Verifier', 'MixedVerifier', 'VerifierError', 'errors', 'wire', 'device_count', 'shard', 'shard_count', 'shard_devices']


def device_count():
    from . import _lib
    return _lib.lib().zkv_device_count()
