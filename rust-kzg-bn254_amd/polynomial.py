"""Mirror of primitives/src/polynomial.rs: `PolynomialEvalForm` / `PolynomialCoeffForm`, a zero-padded
power-of-two vector of Fr plus the byte length of the underlying blob.  `to_coeff_form` / `to_eval_form`
(polynomial.rs:130-140, :241-251) run the HIP NTT through `kzg_fr_ntt`."""
import numpy as np

from . import _lib
from .consts import BYTES_PER_FIELD_ELEMENT, MAINNET_SRS_G1_SIZE
from .errors import GenericError, PolynomialFFTError


def _next_pow2(n: int) -> int:
    p = 1
    while p < n:
        p <<= 1
    return p if n else 0            # usize::next_power_of_two(0) == 1, but an empty Vec stays empty only via resize(1)


def _pad(elems):
    elems = _lib.as_u64(elems, 4).reshape(-1, 4)
    if elems.shape[0] > MAINNET_SRS_G1_SIZE:
        raise GenericError("Input size exceeds maximum polynomial size")
    n = elems.shape[0]
    m = 1
    while m < n:
        m <<= 1                     # next_power_of_two(0) == 1: an empty input becomes [0]
    out = np.zeros((m, 4), dtype=np.uint64)
    out[:n] = elems
    return out, n


def _ntt(data, inverse, what, ctx):
    ctx = ctx or _lib.default_context()
    a = np.ascontiguousarray(data, dtype=np.uint64).copy()
    rc = _lib.load().kzg_fr_ntt(ctx.handle, _lib.ptr(a), a.shape[0], 1 if inverse else 0)
    if rc in (_lib.ERR_DOMAIN, _lib.ERR_NOT_POWER_OF_TWO):
        raise PolynomialFFTError(f"Failed to construct domain for {what}")
    ctx.check_device(rc)
    return a


class PolynomialEvalForm:
    def __init__(self, evals, _blob_len=None):
        self._evaluations, n = _pad(evals)
        self._len_underlying_blob_bytes = n * BYTES_PER_FIELD_ELEMENT if _blob_len is None else _blob_len

    def evaluations(self):
        return self._evaluations

    def __len__(self):
        return self._evaluations.shape[0]

    def len_underlying_blob_bytes(self):
        return self._len_underlying_blob_bytes

    def len_underlying_blob_field_elements(self):
        return self._len_underlying_blob_bytes // BYTES_PER_FIELD_ELEMENT

    def get_evalualtion(self, i):
        return self._evaluations[i] if 0 <= i < len(self) else None

    def is_empty(self):
        return len(self) == 0

    def to_bytes_be(self):
        from .helpers import to_byte_array
        return to_byte_array(self._evaluations, self._len_underlying_blob_bytes)

    def to_coeff_form(self, ctx=None):
        """polynomial.rs:130-140 (IFFT)."""
        return PolynomialCoeffForm(_ntt(self._evaluations, True, "IFFT", ctx), self._len_underlying_blob_bytes)


class PolynomialCoeffForm:
    def __init__(self, coeffs, _blob_len=None):
        self._coeffs, n = _pad(coeffs)
        self._len_underlying_blob_bytes = n * BYTES_PER_FIELD_ELEMENT if _blob_len is None else _blob_len

    def coeffs(self):
        return self._coeffs

    def __len__(self):
        return self._coeffs.shape[0]

    def len_underlying_blob_bytes(self):
        return self._len_underlying_blob_bytes

    def len_underlying_blob_field_elements(self):
        return self._len_underlying_blob_bytes // BYTES_PER_FIELD_ELEMENT

    def get_at_index(self, i):
        return self._coeffs[i] if 0 <= i < len(self) else None

    def is_empty(self):
        return len(self) == 0

    def to_bytes_be(self):
        from .helpers import to_byte_array
        return to_byte_array(self._coeffs, self._len_underlying_blob_bytes)

    def to_eval_form(self, ctx=None):
        """polynomial.rs:241-251 (FFT)."""
        return PolynomialEvalForm(_ntt(self._coeffs, False, "FFT", ctx), self._len_underlying_blob_bytes)
