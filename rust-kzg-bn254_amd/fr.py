"""Wire-format helpers: `Fr` / `Fq` elements as numpy uint64[4] little-endian limbs in Montgomery form
(arkworks' in-memory layout), G1 affine points as uint64[8] = x || y, identity = zeros."""
import numpy as np

from .consts import FQ_MODULUS, FR_MODULUS, MONT_R

_M64 = 0xFFFFFFFFFFFFFFFF
_RINV_R = pow(MONT_R, -1, FR_MODULUS)
_RINV_Q = pow(MONT_R, -1, FQ_MODULUS)


def _limbs(v: int) -> np.ndarray:
    return np.array([(v >> (64 * i)) & _M64 for i in range(4)], dtype=np.uint64)


def _int(a) -> int:
    a = np.asarray(a, dtype=np.uint64).reshape(-1)
    return int(a[0]) | int(a[1]) << 64 | int(a[2]) << 128 | int(a[3]) << 192


def fr_from_int(v: int) -> np.ndarray:
    return _limbs(v % FR_MODULUS * MONT_R % FR_MODULUS)


def fr_to_int(a) -> int:
    return _int(a) * _RINV_R % FR_MODULUS


def frs_from_ints(vals) -> np.ndarray:
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        m = v % FR_MODULUS * MONT_R % FR_MODULUS
        out[i, 0] = m & _M64; out[i, 1] = (m >> 64) & _M64; out[i, 2] = (m >> 128) & _M64; out[i, 3] = m >> 192
    return out


def frs_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [_int(arr[i]) * _RINV_R % FR_MODULUS for i in range(arr.shape[0])]


def fq_from_int(v: int) -> np.ndarray:
    return _limbs(v % FQ_MODULUS * MONT_R % FQ_MODULUS)


def fq_to_int(a) -> int:
    return _int(a) * _RINV_Q % FQ_MODULUS


def g1_from_ints(pt) -> np.ndarray:
    out = np.zeros(8, dtype=np.uint64)
    if pt is not None:
        out[:4] = fq_from_int(pt[0]); out[4:] = fq_from_int(pt[1])
    return out


def g1_to_ints(a):
    a = np.asarray(a, dtype=np.uint64).reshape(8)
    if not a.any():
        return None
    return (fq_to_int(a[:4]), fq_to_int(a[4:]))


def g1_is_identity(a) -> bool:
    return not np.asarray(a).any()
