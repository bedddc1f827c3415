"""Pure-Python big-int reference for BN254 (test infrastructure; small cases only).

Independent of oracle/ (no shared code): plain modular arithmetic and affine group law written from
the curve equation y^2 = x^3 + 3 (reference primitives/src/helpers.rs:202, :244) and the field
moduli of SURVEY.md Appendix A.  Used to pin the C oracle and to convert between decimal fixtures
and the wire format (4 little-endian u64 limbs, Montgomery R = 2^256).
"""
import numpy as np

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # Fq
R_ = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # Fr
MONT_R = 1 << 256
G1 = (1, 2)


# ----- wire-format conversion ---------------------------------------------------------------
def to_limbs(v: int) -> np.ndarray:
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def from_limbs(a) -> int:
    a = np.asarray(a, dtype=np.uint64).reshape(-1)
    return sum(int(a[i]) << (64 * i) for i in range(4))


def fr_to_mont(v: int) -> np.ndarray:
    return to_limbs(v % R_ * MONT_R % R_)


def fr_from_mont(a) -> int:
    return from_limbs(a) * pow(MONT_R, -1, R_) % R_


def fq_to_mont(v: int) -> np.ndarray:
    return to_limbs(v % P * MONT_R % P)


def fq_from_mont(a) -> int:
    return from_limbs(a) * pow(MONT_R, -1, P) % P


def frs_to_mont(vals) -> np.ndarray:
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = fr_to_mont(v)
    return out


def frs_from_mont(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [fr_from_mont(arr[i]) for i in range(arr.shape[0])]


def point_to_wire(pt) -> np.ndarray:
    """affine (x, y) ints or None (identity) -> 8 u64 (x||y Montgomery); identity = zeros."""
    out = np.zeros(8, dtype=np.uint64)
    if pt is not None:
        out[:4] = fq_to_mont(pt[0])
        out[4:] = fq_to_mont(pt[1])
    return out


def point_from_wire(a):
    a = np.asarray(a, dtype=np.uint64).reshape(8)
    if not a.any():
        return None
    return (fq_from_mont(a[:4]), fq_from_mont(a[4:]))


def points_to_wire(pts) -> np.ndarray:
    out = np.zeros((len(pts), 8), dtype=np.uint64)
    for i, p in enumerate(pts):
        out[i] = point_to_wire(p)
    return out


# ----- group law (affine, None = identity) --------------------------------------------------
def ec_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def ec_neg(a):
    return None if a is None else (a[0], (-a[1]) % P)


def ec_mul(k: int, a):
    k %= R_
    acc = None
    while k:
        if k & 1:
            acc = ec_add(acc, a)
        a = ec_add(a, a)
        k >>= 1
    return acc


def on_curve(a) -> bool:
    return a is None or (a[1] * a[1] - a[0] ** 3 - 3) % P == 0


def msm(points, scalars):
    acc = None
    for p, s in zip(points, scalars):
        acc = ec_add(acc, ec_mul(s, p))
    return acc


# ----- domain / transforms ------------------------------------------------------------------
def root_of_unity(log_n: int) -> int:
    return pow(5, (R_ - 1) >> log_n, R_)


def dft(vals, inverse=False):
    """O(n^2) definition of ark-poly's fft/ifft on the domain {w^i}, natural order."""
    n = len(vals)
    w = root_of_unity(n.bit_length() - 1)
    if inverse:
        w = pow(w, -1, R_)
    out = []
    for i in range(n):
        wi = pow(w, i, R_)
        acc, cur = 0, 1
        for j in range(n):
            acc = (acc + vals[j] * cur) % R_
            cur = cur * wi % R_
        out.append(acc)
    if inverse:
        ninv = pow(n, -1, R_)
        out = [v * ninv % R_ for v in out]
    return out


def poly_eval(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R_
    return acc


# ----- blob codec (helpers.rs:823-840, :40-57) ------------------------------------------------
def pad_payload(raw: bytes) -> bytes:
    out = bytearray()
    for s in range(0, len(raw), 31):
        chunk = raw[s:s + 31]
        out += b"\x00" + chunk + b"\x00" * (31 - len(chunk))
    return bytes(out)


def to_fr_array(data: bytes):
    out = []
    for s in range(0, len(data), 32):
        chunk = data[s:s + 32]
        chunk = chunk + b"\x00" * (32 - len(chunk))
        out.append(int.from_bytes(chunk, "big") % R_)
    return out


def next_pow2(n: int) -> int:
    p = 1
    while p < n:
        p <<= 1
    return p
