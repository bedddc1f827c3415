"""Mirror of the reference's `rust-kzg-bn254-verifier` crate (verifier/src/verify.rs, verifier/src/batch.rs).

The O(1) pairing check runs on the host inside the library (csrc/host_pairing.h); the data-parallel parts — barycentric
evaluation of every blob and the three n-point linear combinations of batch verification — run on the GPU through the
same C-ABI as the prover."""
import ctypes as C
import hashlib

import numpy as np

from . import _lib, helpers
from .consts import BYTES_PER_FIELD_ELEMENT, FR_MODULUS, RANDOM_CHALLENGE_KZG_BATCH_DOMAIN
from .errors import GenericError, InvalidInputLength, NotOnCurveError
from .fr import fr_from_int, fr_to_int


def _raise_for(rc, ctx=None):
    if rc == _lib.OK:
        return
    if rc == _lib.ERR_G1_NOT_ON_CURVE:
        raise NotOnCurveError("G1 point not on curve")
    if rc == _lib.ERR_G2_TAU_NOT_ON_CURVE:
        raise NotOnCurveError("Invalid trusted setup: G2_TAU not on curve")
    if rc == _lib.ERR_TAU_EQUALS_Z:
        raise GenericError("Evaluation point equals trusted setup secret")
    if ctx is not None:
        ctx.check_device(rc)
    raise GenericError(_lib.status_message(rc))


def _g2_arg(g2_tau):
    return None if g2_tau is None else _lib.ptr(_lib.as_u64(g2_tau, 0).reshape(16))


def verify_proof(commitment, proof, value_fr, z_fr, g2_tau=None) -> bool:
    """verify.rs:10-72.  `g2_tau=None` uses consts::G2_TAU like the reference; tests with a generated SRS pass their own."""
    ok = _lib.i32(0)
    tau = None if g2_tau is None else _lib.as_u64(g2_tau, 0).reshape(16)
    rc = _lib.load().kzg_verify_proof(_lib.ptr(_lib.as_u64(commitment, 0).reshape(8)), _lib.ptr(_lib.as_u64(proof, 0).reshape(8)),
                                      _lib.ptr(_lib.as_u64(value_fr, 0).reshape(4)), _lib.ptr(_lib.as_u64(z_fr, 0).reshape(4)),
                                      None if tau is None else _lib.ptr(tau), C.byref(ok))
    _raise_for(rc)
    return bool(ok.value)


def verify_blob_kzg_proof(blob, commitment, proof, g2_tau=None, ctx=None) -> bool:
    """verify.rs:76-98 as ONE call of the C-ABI (`kzg_verify_blob_kzg_proof`)."""
    ctx = ctx or _lib.default_context()
    data = np.frombuffer(bytes(blob.data()), dtype=np.uint8) if len(blob.data()) else np.zeros(1, np.uint8)
    ok = _lib.i32(0)
    tau = None if g2_tau is None else _lib.as_u64(g2_tau, 0).reshape(16)
    rc = _lib.load().kzg_verify_blob_kzg_proof(ctx.handle, data.ctypes.data_as(_lib.u8p), len(blob.data()),
                                               _lib.ptr(_lib.as_u64(commitment, 0).reshape(8)), _lib.ptr(_lib.as_u64(proof, 0).reshape(8)),
                                               None if tau is None else _lib.ptr(tau), C.byref(ok))
    ctx.check_device(rc)
    _raise_for(rc)
    return bool(ok.value)


def verify_blob_kzg_proof_composed(blob, commitment, proof, g2_tau=None, ctx=None) -> bool:
    """verify.rs:76-98 composed from the reference's own steps (three calls); the one-call form above must agree with it."""
    helpers.validate_g1_point(commitment)
    helpers.validate_g1_point(proof)
    polynomial = blob.to_polynomial_eval_form()
    z = helpers.compute_challenge(blob, commitment)
    y = helpers.evaluate_polynomial_in_evaluation_form(polynomial, z, ctx)
    return verify_proof(commitment, proof, y, z, g2_tau)


def _pack(items, cols):
    n = len(items)
    return np.ascontiguousarray(np.stack([_lib.as_u64(x, 0).reshape(cols) for x in items])) if n else np.zeros((0, cols), np.uint64)


def compute_r_powers(commitments, zs, ys, proofs, blobs_as_field_elements_length) -> np.ndarray:
    """batch.rs:76-168 (`kzg_compute_r_powers`): r = H(domain || 0^8 || u64be(n) || n x u64be(len_i) || n x (C_i || z_i || y_i || proof_i)),
    returns [r^0 .. r^(n-1)]."""
    n = len(commitments)
    if n == 0:
        return np.zeros((0, 4), dtype=np.uint64)
    if not (len(zs) >= n and len(ys) >= n and len(proofs) >= n and len(blobs_as_field_elements_length) >= n):
        raise InvalidInputLength()
    cm, pf, z_, y_ = _pack(commitments, 8), _pack(proofs[:n], 8), _pack(zs[:n], 4), _pack(ys[:n], 4)
    lens = np.ascontiguousarray([int(v) for v in blobs_as_field_elements_length[:n]], dtype=np.uint64)
    out = np.zeros((n, 4), dtype=np.uint64)
    rc = _lib.load().kzg_compute_r_powers(_lib.ptr(cm), _lib.ptr(z_), _lib.ptr(y_), _lib.ptr(pf), _lib.ptr(lens), n, _lib.ptr(out))
    if rc != _lib.OK:
        raise GenericError(_lib.status_message(rc))
    return out


def compute_r_powers_py(commitments, zs, ys, proofs, blobs_as_field_elements_length) -> np.ndarray:
    """The same transcript assembled in Python (hashlib): an independent cross-check of the C path."""
    n = len(commitments)
    head = bytearray(40)
    head[0:24] = RANDOM_CHALLENGE_KZG_BATCH_DOMAIN
    head[32:40] = helpers.usize_to_be_bytes(n)
    parts = [bytes(head)] + [int(length).to_bytes(8, "big") for length in blobs_as_field_elements_length[:n]]
    for i in range(n):
        parts.append(helpers.serialize_compressed(commitments[i]))
        parts.append(fr_to_int(zs[i]).to_bytes(BYTES_PER_FIELD_ELEMENT, "big"))
        parts.append(fr_to_int(ys[i]).to_bytes(BYTES_PER_FIELD_ELEMENT, "big"))
        parts.append(helpers.serialize_compressed(proofs[i]))
    data = b"".join(parts)
    if len(data) != 40 + n * (4 * BYTES_PER_FIELD_ELEMENT + 8):
        raise InvalidInputLength()
    return helpers.compute_powers(helpers.hash_to_field_element(data), n)


def verify_kzg_proof_batch(commitments, zs, ys, proofs, blobs_as_field_elements_length, g2_tau=None, ctx=None) -> bool:
    """batch.rs:185-256."""
    if not (len(commitments) == len(zs) == len(ys) == len(proofs)):
        raise GenericError("length's of the input are not the same")
    for c in commitments:
        helpers.validate_g1_point(c)
    for p in proofs:
        helpers.validate_g1_point(p)
    ctx = ctx or _lib.default_context()
    n = len(commitments)
    r_powers = compute_r_powers(commitments, zs, ys, proofs, blobs_as_field_elements_length)

    cm, pf, z_, y_ = _pack(commitments, 8), _pack(proofs, 8), _pack(zs, 4), _pack(ys, 4)
    rp = np.ascontiguousarray(_lib.as_u64(r_powers, 4).reshape(-1, 4))
    tau = None if g2_tau is None else _lib.as_u64(g2_tau, 0).reshape(16)
    ok = _lib.i32(0)
    nul = None
    rc = _lib.load().kzg_verify_kzg_proof_batch(ctx.handle, _lib.ptr(cm) if n else nul, _lib.ptr(z_) if n else nul, _lib.ptr(y_) if n else nul,
                                                _lib.ptr(pf) if n else nul, _lib.ptr(rp) if n else nul, n,
                                                None if tau is None else _lib.ptr(tau), C.byref(ok))
    _raise_for(rc, ctx)
    return bool(ok.value)


def verify_blob_kzg_proof_batch(blobs, commitments, proofs, g2_tau=None, ctx=None) -> bool:
    """batch.rs:16-69 as ONE call of the C-ABI (`kzg_verify_blob_kzg_proof_batch`): point validation, the n Fiat-Shamir challenges
    (host thread pool), the n barycentric evaluations (one batched GPU launch), compute_r_powers, three batched GPU MSMs and the
    host 2-pairing check."""
    if not (len(commitments) == len(blobs) and len(proofs) == len(blobs)):
        raise GenericError("length's of the input are not the same")
    ctx = ctx or _lib.default_context()
    n = len(blobs)
    ptrs, lens, _keep = _lib.blob_args(blobs)
    cm, pf = _pack(commitments, 8), _pack(proofs, 8)
    tau = None if g2_tau is None else _lib.as_u64(g2_tau, 0).reshape(16)
    ok = _lib.i32(0)
    rc = _lib.load().kzg_verify_blob_kzg_proof_batch(ctx.handle, ptrs if n else None, lens if n else None, _lib.ptr(cm) if n else None,
                                                     _lib.ptr(pf) if n else None, n, None if tau is None else _lib.ptr(tau), C.byref(ok))
    _raise_for(rc, ctx)
    return bool(ok.value)


def verify_blob_kzg_proof_batch_py(blobs, commitments, proofs, g2_tau=None, ctx=None) -> bool:
    """The same flow step by step through the separate entry points (cross-check of the one-call form)."""
    if not (len(commitments) == len(blobs) and len(proofs) == len(blobs)):
        raise GenericError("length's of the input are not the same")
    for c in commitments:
        helpers.validate_g1_point(c)
    for p in proofs:
        helpers.validate_g1_point(p)
    zs, ys = helpers.compute_challenges_and_evaluate_polynomial(blobs, commitments, ctx)
    lengths = [len(b.to_polynomial_eval_form()) for b in blobs]
    return verify_kzg_proof_batch(commitments, zs, ys, proofs, lengths, g2_tau, ctx)
