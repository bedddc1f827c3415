"""ctypes binding of oracle/liboracle.so — the CPU parity checker (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "liboracle.so")
_lib = None

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_init()
        _lib.orc_calculate_roots_of_unity.restype = C.c_long
        _lib.orc_pad_payload.restype = C.c_size_t
        _lib.orc_to_fr_array.restype = C.c_size_t
        _lib.orc_ark_window.restype = C.c_uint
    return _lib


def _p(a):
    return a.ctypes.data_as(u64p)


def _b(a):
    return a.ctypes.data_as(u8p)


def _u64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a if shape is None else a.reshape(shape)


FQ, FR = 0, 1


def constants(which):
    m = np.zeros(4, np.uint64); one = np.zeros(4, np.uint64); r2 = np.zeros(4, np.uint64)
    inv = C.c_uint64()
    lib().orc_constants(which, _p(m), C.byref(inv), _p(one), _p(r2))
    return m, inv.value, one, r2


def f_mul(which, a, b):
    r = np.zeros(4, np.uint64); a = _u64(a); b = _u64(b)
    lib().orc_f_mul(which, _p(r), _p(a), _p(b)); return r


def f_add(which, a, b):
    r = np.zeros(4, np.uint64); a = _u64(a); b = _u64(b)
    lib().orc_f_add(which, _p(r), _p(a), _p(b)); return r


def f_sub(which, a, b):
    r = np.zeros(4, np.uint64); a = _u64(a); b = _u64(b)
    lib().orc_f_sub(which, _p(r), _p(a), _p(b)); return r


def f_inv(which, a):
    r = np.zeros(4, np.uint64); a = _u64(a)
    ok = lib().orc_f_inv(which, _p(r), _p(a)); return r if ok else None


def f_from_be_bytes_mod_order(which, data: bytes):
    r = np.zeros(4, np.uint64); buf = np.frombuffer(data, dtype=np.uint8).copy()
    lib().orc_f_from_be_bytes_mod_order(which, _p(r), _b(buf), C.c_size_t(len(data))); return r


def f_to_be_bytes(which, a) -> bytes:
    out = np.zeros(32, np.uint8); a = _u64(a)
    lib().orc_f_to_be_bytes(which, _b(out), _p(a)); return out.tobytes()


def montgomery_reduce(limbs):
    a = _u64(limbs); out = np.zeros(4, np.uint64)
    lib().orc_montgomery_reduce(_p(a), _p(out)); return out


def g1_decompress_be(data: bytes):
    buf = np.frombuffer(data, dtype=np.uint8).copy(); out = np.zeros(8, np.uint64)
    rc = lib().orc_g1_decompress_be(_b(buf), _p(out)); return rc, out


def g1_serialize_compressed_ark(xy) -> bytes:
    xy = _u64(xy); out = np.zeros(32, np.uint8)
    lib().orc_g1_serialize_compressed_ark(_p(xy), _b(out)); return out.tobytes()


def g1_is_on_curve(xy) -> bool:
    xy = _u64(xy); return bool(lib().orc_g1_is_on_curve(_p(xy)))


def g1_add(a, b):
    a = _u64(a); b = _u64(b); out = np.zeros(8, np.uint64)
    lib().orc_g1_add(_p(out), _p(a), _p(b)); return out


def g1_neg(a):
    a = _u64(a); out = np.zeros(8, np.uint64)
    lib().orc_g1_neg(_p(out), _p(a)); return out


def g1_scalar_mul(p, k_mont):
    p = _u64(p); k = _u64(k_mont); out = np.zeros(8, np.uint64)
    lib().orc_g1_scalar_mul(_p(out), _p(p), _p(k)); return out


def g1_jacobian_to_affine(xyz):
    xyz = _u64(xyz); out = np.zeros(8, np.uint64)
    lib().orc_g1_jacobian_to_affine(_p(out), _p(xyz)); return out


def msm_naive(bases, scalars):
    bases = _u64(bases, (-1, 8)); scalars = _u64(scalars, (-1, 4)); out = np.zeros(8, np.uint64)
    lib().orc_msm_naive(_p(bases), _p(scalars), C.c_size_t(len(scalars)), _p(out)); return out


def host_cpus():
    """CPUs this process may really use: hardware threads, capped by its affinity mask and by the cgroup's CFS quota (a GPU box shows 256 hardware
    threads under a 16-CPU quota: 256 oracle threads there are throttled by the kernel and run SLOWER than 16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            n = min(n, max(1, int(float(a) / float(b))))
    except (OSError, ValueError):
        pass
    return n


def msm_pippenger(bases, scalars, threads=None):
    bases = _u64(bases, (-1, 8)); scalars = _u64(scalars, (-1, 4)); out = np.zeros(8, np.uint64)
    n = min(len(bases), len(scalars))
    lib().orc_msm_pippenger(_p(bases), _p(scalars), C.c_size_t(n), _p(out), threads or os.cpu_count() or 1)
    return out


def ark_window(n):
    return lib().orc_ark_window(C.c_size_t(n))


def fr_ntt(data, inverse=False):
    a = _u64(data, (-1, 4)).copy()
    rc = lib().orc_fr_ntt(_p(a), C.c_size_t(len(a)), int(inverse))
    if rc:
        raise ValueError("oracle NTT: length is not a power of two <= 2^28")
    return a


def fr_ntt_mt(data, inverse=False, threads=None):
    """The same transform, every layer chunked over `threads` pthreads (CPU baseline of the NTT)."""
    a = _u64(data, (-1, 4)).copy()
    rc = lib().orc_fr_ntt_mt(_p(a), C.c_size_t(len(a)), int(inverse), int(threads or host_cpus()))
    if rc:
        raise ValueError("oracle NTT: length is not a power of two <= 2^28")
    return a


def fr_root_of_unity(log_n):
    out = np.zeros(4, np.uint64); lib().orc_fr_root_of_unity(log_n, _p(out)); return out


def g1_ifft(points, n):
    points = _u64(points, (-1, 8)); out = np.zeros((n, 8), np.uint64)
    rc = lib().orc_g1_ifft(_p(points), C.c_size_t(n), _p(out)); return rc, out


def calculate_roots_of_unity(len_bytes, cap=None):
    elems = (len_bytes + 31) // 32
    n = 1
    while n < elems:
        n <<= 1
    cap = cap or max(n, 1)
    out = np.zeros((cap, 4), np.uint64)
    rc = lib().orc_calculate_roots_of_unity(C.c_uint64(len_bytes), _p(out), C.c_size_t(cap))
    return rc, out[:max(rc, 0)]


def pad_payload(raw: bytes) -> bytes:
    buf = np.frombuffer(raw, dtype=np.uint8).copy() if raw else np.zeros(1, np.uint8)
    out = np.zeros(((len(raw) + 30) // 31) * 32 + 32, np.uint8)
    n = lib().orc_pad_payload(_b(buf), C.c_size_t(len(raw)), _b(out)); return out[:n].tobytes()


def to_fr_array(data: bytes):
    buf = np.frombuffer(data, dtype=np.uint8).copy() if data else np.zeros(1, np.uint8)
    n = (len(data) + 31) // 32
    out = np.zeros((max(n, 1), 4), np.uint64)
    lib().orc_to_fr_array(_b(buf), C.c_size_t(len(data)), _p(out)); return out[:n]


def evaluate_polynomial_in_evaluation_form(evals, z):
    evals = _u64(evals, (-1, 4)); z = _u64(z); out = np.zeros(4, np.uint64)
    rc = lib().orc_evaluate_polynomial_in_evaluation_form(_p(evals), C.c_size_t(len(evals)), _p(z), _p(out))
    return rc, out


def commit_coeff_form(srs, coeffs, threads=None):
    srs = _u64(srs, (-1, 8)); coeffs = _u64(coeffs, (-1, 4)); out = np.zeros(8, np.uint64)
    rc = lib().orc_commit_coeff_form(_p(srs), C.c_size_t(len(srs)), _p(coeffs), C.c_size_t(len(coeffs)), _p(out),
                                     threads or os.cpu_count() or 1)
    return rc, out


def commit_eval_form(srs, evals, literal=True, threads=None):
    srs = _u64(srs, (-1, 8)); evals = _u64(evals, (-1, 4)); out = np.zeros(8, np.uint64)
    fn = lib().orc_commit_eval_form if literal else lib().orc_commit_eval_form_via_ifft
    rc = fn(_p(srs), C.c_size_t(len(srs)), _p(evals), C.c_size_t(len(evals)), _p(out), threads or os.cpu_count() or 1)
    return rc, out


def compute_proof(srs, evals, roots, z, literal=True, threads=None, want_quotient=False):
    srs = _u64(srs, (-1, 8)); evals = _u64(evals, (-1, 4)); roots = _u64(roots, (-1, 4)); z = _u64(z)
    out = np.zeros(8, np.uint64); y = np.zeros(4, np.uint64)
    q = np.zeros((len(evals), 4), np.uint64) if want_quotient else None
    rc = lib().orc_compute_proof(_p(srs), C.c_size_t(len(srs)), _p(evals), C.c_size_t(len(evals)),
                                 _p(roots), C.c_size_t(len(roots)), _p(z), _p(out), _p(y),
                                 _p(q) if want_quotient else None, int(literal), threads or os.cpu_count() or 1)
    return (rc, out, y, q) if want_quotient else (rc, out, y)


def compute_challenge(blob: bytes, commitment_xy):
    buf = np.frombuffer(blob, dtype=np.uint8).copy(); c = _u64(commitment_xy); z = np.zeros(4, np.uint64)
    lib().orc_compute_challenge(_b(buf), C.c_size_t(len(blob)), _p(c), _p(z)); return z


def sha256(msg: bytes) -> bytes:
    buf = np.frombuffer(msg, dtype=np.uint8).copy() if msg else np.zeros(1, np.uint8)
    out = np.zeros(32, np.uint8)
    lib().orc_sha256(_b(buf), C.c_size_t(len(msg)), _b(out)); return out.tobytes()


def compute_challenges_and_evaluate_polynomial(blobs, commitments):
    """helpers.rs:613-662; `blobs` = list of padded blob bytes, commitments (n, 8)."""
    n = len(blobs)
    packed = np.frombuffer(b"".join(blobs) or b"\0", dtype=np.uint8).copy()
    lens = np.array([len(b) for b in blobs], dtype=np.uint64)
    cm = _u64(commitments, (-1, 8)); zs = np.zeros((n, 4), np.uint64); ys = np.zeros((n, 4), np.uint64)
    rc = lib().orc_compute_challenges_and_evaluate_polynomial(_b(packed), _p(lens), _p(cm), C.c_size_t(n), _p(zs), _p(ys))
    return rc, zs, ys


def compute_r_powers(commitments, zs, ys, proofs, lens):
    """verifier/src/batch.rs:76-168."""
    cm = _u64(commitments, (-1, 8)); pf = _u64(proofs, (-1, 8)); z = _u64(zs, (-1, 4)); y = _u64(ys, (-1, 4))
    n = len(cm); ln = np.array([int(v) for v in lens], dtype=np.uint64); out = np.zeros((n, 4), np.uint64)
    lib().orc_compute_r_powers(_p(cm), _p(z), _p(y), _p(pf), _p(ln), C.c_size_t(n), _p(out))
    return out
