"""halo2_rsa_amd -- MI355X-native witness engine for the halo2-rsa BigIntChip/RSAChip hot path.

`halo2_rsa_amd/csrc` holds the hand-written HIP kernels and the C ABI (include/h2r.h);
`big_integer` / `rsa` mirror the reference's chip API for that path.  No CPU fallback exists.
"""
from ._lib import (H2R_E_NOT_IN_FIELD, H2R_E_NOT_REDUCED, H2R_E_SHAPE, H2R_E_ZERO_MODULUS, H2R_OK, H2RError,  # noqa: F401
                   lib, lib_path)
from .big_integer import (AssignedInteger, BatchResult, BigIntChip, LookupArgument, Pipeline, Trace, TraceArena,  # noqa: F401
                          UnassignedInteger)
from .rsa import (Fix, RSAChip, RSAPublicKey, RSASignature, RSASignatureVerifier, Var, hashed_msg_from_digest,  # noqa: F401
                  pack_messages, sha256_hashed_msg, signature_from_bytes_be)
