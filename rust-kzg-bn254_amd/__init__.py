"""rust-kzg-bn254_amd — MI355X-native (gfx950) KZG-BN254 prover hot path behind the reference's API.

Python host mirror of Layr-Labs/rust-kzg-bn254's prover surface (`KZG`, `SRS`, `Blob`,
`PolynomialEvalForm` / `PolynomialCoeffForm`, the path-relevant `helpers`), over the C-ABI of
include/kzg_bn254_mi355x.h (libkzg_bn254_mi355x.so, hand-written HIP kernels in csrc/).
Import as `rust_kzg_bn254_amd` (shim package at the repo root).
"""
from . import consts, errors, fr, helpers, sharding, verifier  # noqa: F401
from ._lib import Context, default_context, load  # noqa: F401
from .blob import Blob  # noqa: F401
from .kzg import KZG  # noqa: F401
from .polynomial import PolynomialCoeffForm, PolynomialEvalForm  # noqa: F401
from .srs import SRS  # noqa: F401
from .verifier import verify_blob_kzg_proof, verify_blob_kzg_proof_batch, verify_proof  # noqa: F401
