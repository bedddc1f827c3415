"""Host-side mirror of the reference's `primitives::helpers` functions that sit on the prover path
(primitives/src/helpers.rs).  Byte codecs and the Fiat-Shamir transcript are O(n) host work (out of
the kernel scope); everything that touches group or NTT arithmetic goes through
the C-ABI to the HIP kernels."""
import ctypes as C
import hashlib
import re

import numpy as np

from . import _lib
from .consts import (BYTES_PER_FIELD_ELEMENT, FIAT_SHAMIR_PROTOCOL_DOMAIN, FQ_MODULUS, FR_MODULUS,
                     MAINNET_SRS_G1_SIZE, primitive_root_of_unity)
from .errors import (DeserializationError, G2GeneratorNotAcceptedError, GenericError, InvalidFieldElement, InvalidInputLength, MsmError,
                     NotOnCurveError)
from .fr import fq_to_int, fr_from_int, fr_to_int, frs_from_ints, frs_to_ints, g1_is_identity


def get_num_element(data_len: int, symbol_size: int) -> int:
    return -(-data_len // symbol_size)


def pad_payload(input_data: bytes) -> bytes:
    """helpers.rs:823-840: a 0x00 in front of every 31-byte chunk, output aligned to 32 bytes."""
    n = len(input_data)
    chunks = -(-n // 31)
    src = np.zeros(chunks * 31, dtype=np.uint8)
    src[:n] = np.frombuffer(input_data, dtype=np.uint8)
    out = np.zeros((chunks, 32), dtype=np.uint8)
    out[:, 1:] = src.reshape(chunks, 31)
    return out.tobytes()


def remove_internal_padding(padded: bytes) -> bytes:
    """helpers.rs:856-874."""
    if len(padded) % BYTES_PER_FIELD_ELEMENT != 0:
        raise InvalidInputLength()
    a = np.frombuffer(padded, dtype=np.uint8).reshape(-1, 32)
    return a[:, 1:].tobytes()


def to_fr_array(data: bytes, ctx=None) -> np.ndarray:
    """helpers.rs:40-57: each 32-byte big-endian chunk (last one right-padded with zeros) mod r.
    Small inputs are converted on the host (pure byte arithmetic, no GPU needed); large ones by `kzg_blob_to_fr`."""
    n = get_num_element(len(data), BYTES_PER_FIELD_ELEMENT)
    if n >= 4096:
        ctx = ctx or _lib.default_context()
        buf = np.frombuffer(data, dtype=np.uint8)
        n_out = C.c_size_t(0)
        lib = _lib.load()
        lib.kzg_blob_to_fr(ctx.handle, None, len(data), None, 0, C.byref(n_out))
        out = np.zeros((n_out.value, 4), dtype=np.uint64)
        rc = lib.kzg_blob_to_fr(ctx.handle, buf.ctypes.data_as(_lib.u8p), len(data), _lib.ptr(out), out.shape[0], C.byref(n_out))
        ctx.check_device(rc)
        return out[:n]
    data = data + b"\x00" * (n * 32 - len(data))
    return frs_from_ints([int.from_bytes(data[32 * i:32 * i + 32], "big") for i in range(n)])


def blob_to_polynomial(blob: bytes, ctx=None) -> np.ndarray:
    """helpers.rs:28-30."""
    return to_fr_array(blob, ctx)


def set_bytes_canonical(data: bytes) -> np.ndarray:
    """helpers.rs:32-34: Fr::from_be_bytes_mod_order."""
    return fr_from_int(int.from_bytes(bytes(data), "big") % FR_MODULUS)


def is_zeroed(first_byte: int, buf) -> bool:
    """helpers.rs:121-132."""
    return first_byte == 0 and not any(bytes(buf))


def str_vec_to_fr_vec(input) -> np.ndarray:
    """helpers.rs:134-149: decimal strings ("-1" spelled out by the reference); ark-ff 0.5's Fr::from_str parses a signed decimal integer and
    reduces it mod r; what it refuses is the reference's panic ("could not load string to Fr")."""
    out = []
    for element in input:
        if not re.fullmatch(r"[+-]?[0-9]+", element):
            raise ValueError("could not load string to Fr")
        out.append(int(element) % FR_MODULUS)
    return frs_from_ints(out)


def validate_blob_data_as_canonical_field_elements(data: bytes) -> None:
    """helpers.rs:784-810: every 32-byte big-endian chunk < r (one vectorised comparison of the four 64-bit words, no loop over the elements)."""
    if len(data) % BYTES_PER_FIELD_ELEMENT != 0:
        raise InvalidInputLength()
    if not data:
        return
    a = np.frombuffer(data, dtype=">u8").reshape(-1, 4)
    m = [(FR_MODULUS >> (64 * (3 - k))) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    ge = a[:, 3] >= m[3]
    for k in (2, 1, 0):
        ge = (a[:, k] > m[k]) | ((a[:, k] == m[k]) & ge)
    if ge.any():
        raise InvalidFieldElement(f"Field element at position {int(np.argmax(ge))} is not canonical or invalid")


def read_g1_point_from_bytes_be(g1_bytes_be: bytes, ctx=None) -> np.ndarray:
    """helpers.rs:175-227: one gnark-compressed point -> wire-format affine point, through the kernel that decodes a whole SRS file
    (`kzg_srs_load_compressed_be`, the square root on the GPU) and a read-back."""
    if len(g1_bytes_be) != 32:
        raise DeserializationError("not enough bytes for g1 point")
    ctx = ctx or _lib.default_context()
    lib = _lib.load()
    h = C.c_void_p()
    bad = C.c_uint64(0)
    buf = np.frombuffer(bytes(g1_bytes_be), dtype=np.uint8)
    rc = lib.kzg_srs_load_compressed_be(ctx.handle, buf.ctypes.data_as(_lib.u8p), 1, C.byref(h), C.byref(bad))
    if rc == _lib.ERR_DESERIALIZE:
        raise DeserializationError("point at infinity not coded properly for g1")
    if rc == _lib.ERR_NOT_ON_CURVE:
        raise NotOnCurveError(f"compressed g1 point not on curve: {list(bytes(g1_bytes_be))}")
    ctx.check_device(rc)
    out = np.zeros(8, dtype=np.uint64)
    try:
        rc = lib.kzg_srs_download(ctx.handle, h, 0, 1, _lib.ptr(out))
        ctx.check_device(rc)
    finally:
        lib.kzg_srs_free(h)
    return out


def to_byte_array(data_fr, max_output_size: int) -> bytes:
    """helpers.rs:80-119."""
    vals = frs_to_ints(data_fr)
    out = b"".join(v.to_bytes(32, "big") for v in vals)
    return out[:min(len(vals) * 32, max_output_size)]


def get_primitive_root_of_unity(power: int) -> np.ndarray:
    try:
        return fr_from_int(primitive_root_of_unity(power))
    except IndexError:
        raise GenericError("power must be <= 28")


def hash_to_field_element(msg: bytes) -> np.ndarray:
    """helpers.rs:382-390: SHA-256, big-endian, mod r."""
    return fr_from_int(int.from_bytes(hashlib.sha256(msg).digest(), "big"))


def calculate_roots_of_unity(length_of_data_after_padding: int, ctx=None) -> np.ndarray:
    """helpers.rs:553-589 -> [1, w, ..., w^(n-1)], n = next_pow2(ceil(len / 32)); generated on the GPU."""
    ctx = ctx or _lib.default_context()
    lib = _lib.load()
    n_out = C.c_size_t(0)
    rc = lib.kzg_calculate_roots_of_unity(ctx.handle, length_of_data_after_padding, None, 0, C.byref(n_out))
    if rc == _lib.ERR_ZERO_LENGTH:
        raise GenericError("Length of data after padding is 0")
    if rc == _lib.ERR_SRS_LENGTH:
        raise GenericError("the length of data after padding is not valid with respect to the SRS")
    out = np.zeros((n_out.value, 4), dtype=np.uint64)
    rc = lib.kzg_calculate_roots_of_unity(ctx.handle, length_of_data_after_padding, _lib.ptr(out), out.shape[0], C.byref(n_out))
    ctx.check_device(rc)
    return out


def evaluate_polynomial_in_evaluation_form(polynomial, z, ctx=None) -> np.ndarray:
    """helpers.rs:475-535 (barycentric; returns the stored evaluation when z is a domain element)."""
    ctx = ctx or _lib.default_context()
    blob_size = polynomial.len_underlying_blob_bytes()
    if blob_size == 0:
        raise GenericError("Length of data after padding is 0")
    elems = -(-blob_size // 32)
    if elems > MAINNET_SRS_G1_SIZE:
        raise GenericError("the length of data after padding is not valid with respect to the SRS")
    n_roots = 1
    while n_roots < elems:
        n_roots <<= 1
    if len(polynomial) != n_roots:
        raise InvalidInputLength()
    evals = _lib.as_u64(polynomial.evaluations(), 4)
    zz = _lib.as_u64(z, 0).reshape(4)
    y = np.zeros(4, dtype=np.uint64)
    rc = _lib.load().kzg_evaluate_polynomial_in_evaluation_form(ctx.handle, _lib.ptr(evals), len(evals), _lib.ptr(zz), _lib.ptr(y))
    ctx.check_device(rc)
    if rc != _lib.OK:
        raise InvalidInputLength()
    return y


def g1_lincomb(points, scalars, ctx=None) -> np.ndarray:
    """helpers.rs:328-337: MSM over caller-provided bases (e.g. batch verification, verifier/src/batch.rs:228-246)."""
    ctx = ctx or _lib.default_context()
    pts = _lib.as_u64(points, 8)
    sc = _lib.as_u64(scalars, 4)
    out = np.zeros(8, dtype=np.uint64)
    inf = C.c_uint8(0)
    rc = _lib.load().kzg_msm_g1(ctx.handle, _lib.ptr(pts), len(pts), _lib.ptr(sc), len(sc), _lib.ptr(out), C.byref(inf))
    if rc == _lib.ERR_MSM_LENGTH_MISMATCH:
        raise MsmError(str(min(len(pts), len(sc))))
    ctx.check_device(rc)
    return out


def g1_lincomb_batch(points_list, scalars_list, ctx=None):
    """Several `g1_lincomb` calls of equal length in one kernel sequence (`kzg_msm_g1_batch`), e.g. the three MSMs of
    verifier/src/batch.rs:228,245,246.  Returns a list of affine wire points."""
    ctx = ctx or _lib.default_context()
    k = len(points_list)
    pts = [_lib.as_u64(p, 8).reshape(-1, 8) for p in points_list]
    scs = [_lib.as_u64(s_, 4).reshape(-1, 4) for s_ in scalars_list]
    n = pts[0].shape[0]
    for p_, s_ in zip(pts, scs):
        if p_.shape[0] != n or s_.shape[0] != n:
            raise MsmError(str(min(p_.shape[0], s_.shape[0])))
    allp = np.ascontiguousarray(np.concatenate(pts)) if n else np.zeros((0, 8), np.uint64)
    alls = np.ascontiguousarray(np.concatenate(scs)) if n else np.zeros((0, 4), np.uint64)
    out = np.zeros((k, 8), dtype=np.uint64)
    inf = np.zeros(k, dtype=np.uint8)
    rc = _lib.load().kzg_msm_g1_batch(ctx.handle, _lib.ptr(allp) if n else None, _lib.ptr(alls) if n else None, n, k, _lib.ptr(out),
                                      inf.ctypes.data_as(_lib.u8p))
    ctx.check_device(rc)
    if rc != _lib.OK:
        raise GenericError(_lib.status_message(rc))
    return [out[i] for i in range(k)]


def lexicographically_largest(y_mont) -> bool:
    """helpers.rs:151-173."""
    return fq_to_int(y_mont) > (FQ_MODULUS - 1) // 2


def serialize_compressed(point) -> bytes:
    """ark-serialize compressed G1Affine as used at helpers.rs:456-459: x little-endian, bit 7 of the last byte
    = y is the lexicographically larger root, bit 6 = infinity."""
    if g1_is_identity(point):
        return bytes(31) + b"\x40"
    p = np.asarray(point, dtype=np.uint64).reshape(8)
    b = bytearray(fq_to_int(p[:4]).to_bytes(32, "little"))
    if lexicographically_largest(p[4:]):
        b[31] |= 0x80
    return bytes(b)


def is_on_curve_g1(point) -> bool:
    """helpers.rs:239-261 (y^2 = x^3 + 3; the identity counts as on the curve)."""
    if g1_is_identity(point):
        return True
    p = np.asarray(point, dtype=np.uint64).reshape(8)
    x, y = fq_to_int(p[:4]), fq_to_int(p[4:])
    return (y * y - x * x * x - 3) % FQ_MODULUS == 0


def validate_g1_point(point) -> None:
    """helpers.rs:694-708: on curve + correct subgroup (G1 of BN254 has cofactor 1, so the second check is implied)."""
    if not is_on_curve_g1(point):
        raise NotOnCurveError("G1 point not on curve")


def is_on_curve_g2(point) -> bool:
    """helpers.rs:263-285 (the twist y^2 = x^3 + 3 / (9 + u)); `point`: G2 wire format, 16 u64."""
    ok = _lib.i32(0)
    rc = _lib.load().kzg_g2_is_on_curve(_lib.ptr(_lib.as_u64(point, 0).reshape(16)), C.byref(ok))
    if rc != _lib.OK:
        raise ValueError(_lib.status_message(rc))
    return bool(ok.value)


def example_validate_g2_point(point) -> None:
    """helpers.rs:740-766: on the curve, not the identity, in the order-r subgroup, not the generator -- in that order."""
    reason = _lib.i32(0)
    rc = _lib.load().kzg_validate_g2_point(_lib.ptr(_lib.as_u64(point, 0).reshape(16)), C.byref(reason))
    if rc != _lib.OK:
        raise ValueError(_lib.status_message(rc))
    if reason.value == 1:
        raise NotOnCurveError("G2 point not on curve")
    if reason.value == 2:
        raise NotOnCurveError("G2 point is point at infinity")
    if reason.value == 3:
        raise NotOnCurveError("G2 point not in correct subgroup")
    if reason.value == 4:
        raise G2GeneratorNotAcceptedError("G2 point cannot be the generator point")


def compute_powers(base, count: int) -> np.ndarray:
    """helpers.rs:298-315: [base^0 .. base^(count-1)]."""
    b = fr_to_int(base)
    out, cur = [], 1
    for _ in range(count):
        out.append(cur)
        cur = cur * b % FR_MODULUS
    return frs_from_ints(out)


def usize_to_be_bytes(number: int) -> bytes:
    """helpers.rs:769-771."""
    return int(number).to_bytes(8, "big")


def pairings_verify(a1, a2, b1, b2) -> bool:
    """helpers.rs:392-398: e(a1, a2) == e(b1, b2).  a2, b2: G2 wire points (16 u64).  O(1) host pairing inside the library."""
    ok = _lib.i32(0)
    rc = _lib.load().kzg_pairings_verify(_lib.ptr(_lib.as_u64(a1, 0).reshape(8)), _lib.ptr(_lib.as_u64(a2, 0).reshape(16)),
                                         _lib.ptr(_lib.as_u64(b1, 0).reshape(8)), _lib.ptr(_lib.as_u64(b2, 0).reshape(16)),
                                         C.byref(ok))
    if rc == _lib.ERR_G1_NOT_ON_CURVE:
        raise NotOnCurveError("G1 point not on curve")
    if rc != _lib.OK:
        raise ValueError(_lib.status_message(rc))
    return bool(ok.value)


def g2_generator() -> np.ndarray:
    out = np.zeros(16, dtype=np.uint64)
    _lib.load().kzg_g2_generator(_lib.ptr(out))
    return out


def g2_tau() -> np.ndarray:
    """consts::G2_TAU (primitives/src/consts.rs:55-64)."""
    out = np.zeros(16, dtype=np.uint64)
    _lib.load().kzg_g2_tau_mainnet(_lib.ptr(out))
    return out


def g2_mul_generator(scalar) -> np.ndarray:
    out = np.zeros(16, dtype=np.uint64)
    _lib.load().kzg_g2_mul_generator(_lib.ptr(_lib.as_u64(scalar, 0).reshape(4)), _lib.ptr(out))
    return out


def compute_challenges_and_evaluate_polynomial(blobs, commitments, ctx=None):
    """helpers.rs:613-665 in ONE call of the C-ABI (`kzg_compute_challenges_and_evaluate_polynomial`): the n transcripts are hashed
    on a pool of host threads, the n barycentric evaluations are one batched GPU launch."""
    if len(blobs) != len(commitments) and len(blobs) != 0:
        raise GenericError("length's of the input are not the same or is empty")
    n = len(blobs)
    if n == 0:
        return [], []
    ctx = ctx or _lib.default_context()
    ptrs, lens, _keep = _lib.blob_args(blobs)
    cm = np.ascontiguousarray(np.stack([_lib.as_u64(c, 0).reshape(8) for c in commitments[:n]]))
    zs = np.zeros((n, 4), dtype=np.uint64)
    ys = np.zeros((n, 4), dtype=np.uint64)
    rc = _lib.load().kzg_compute_challenges_and_evaluate_polynomial(ctx.handle, ptrs, lens, _lib.ptr(cm), n, _lib.ptr(zs), _lib.ptr(ys))
    if rc == _lib.ERR_G1_NOT_ON_CURVE:
        raise NotOnCurveError("G1 point not on curve")
    ctx.check_device(rc)
    if rc != _lib.OK:
        raise GenericError(_lib.status_message(rc))
    return list(zs), list(ys)


def evaluate_blobs_in_evaluation_form_batch(blobs, zs, ctx=None):
    """y_i = p_i(z_i) for n blobs in one batched GPU launch (`kzg_evaluate_blobs_in_evaluation_form_batch`; helpers.rs:475-535 each)."""
    n = len(blobs)
    if n == 0:
        return []
    ctx = ctx or _lib.default_context()
    ptrs, lens, _keep = _lib.blob_args(blobs)
    z = np.ascontiguousarray(np.stack([_lib.as_u64(v, 0).reshape(4) for v in zs[:n]]))
    ys = np.zeros((n, 4), dtype=np.uint64)
    rc = _lib.load().kzg_evaluate_blobs_in_evaluation_form_batch(ctx.handle, ptrs, lens, _lib.ptr(z), n, _lib.ptr(ys))
    ctx.check_device(rc)
    if rc != _lib.OK:
        raise GenericError(_lib.status_message(rc))
    return list(ys)


def compute_challenges_and_evaluate_polynomial_py(blobs, commitments, ctx=None):
    """The same, blob by blob through the single-blob entry points: kept as a cross-check of the batched call."""
    if len(blobs) != len(commitments) and len(blobs) != 0:
        raise GenericError("length's of the input are not the same or is empty")
    zs, ys = [], []
    for blob, commitment in zip(blobs, commitments):
        poly = blob.to_polynomial_eval_form()
        z = compute_challenge(blob, commitment)
        ys.append(evaluate_polynomial_in_evaluation_form(poly, z, ctx))
        zs.append(z)
    return zs, ys


def compute_challenge(blob, commitment) -> np.ndarray:
    """helpers.rs:411-472: SHA-256(tag || u64be(n) || n x 32-byte evaluations || compressed commitment) mod r
    (`kzg_compute_challenge`: the transcript is built and hashed in C on the host, no Python loop over the elements)."""
    data = blob.data()
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    z = np.zeros(4, dtype=np.uint64)
    rc = _lib.load().kzg_compute_challenge(buf.ctypes.data_as(_lib.u8p), len(data), _lib.ptr(_lib.as_u64(commitment, 0).reshape(8)), _lib.ptr(z))
    if rc == _lib.ERR_G1_NOT_ON_CURVE:
        raise NotOnCurveError("G1 point not on curve")
    if rc != _lib.OK:
        raise GenericError(_lib.status_message(rc))
    return z


def compute_challenge_py(blob, commitment) -> np.ndarray:
    """The same transcript assembled in Python (hashlib): kept as an independent cross-check of the C path."""
    validate_g1_point(commitment)
    poly = blob.to_polynomial_eval_form()
    n = len(poly)
    data = to_byte_array(poly.evaluations(), n * BYTES_PER_FIELD_ELEMENT)
    msg = FIAT_SHAMIR_PROTOCOL_DOMAIN + n.to_bytes(8, "big") + data + serialize_compressed(commitment)
    return hash_to_field_element(msg)
