"""-m gpu: the reference's own LIVE tests, one for one and under their own names, through the Python mirror of its API (which calls the
C-ABI / the HIP kernels for every commitment, transform, evaluation and proof).  Files mirrored:

    prover/tests/kzg_test.rs            (all 6 tests)
    primitives/tests/polynomial_test.rs (all 5)
    primitives/tests/blob_test.rs       (all 4; the 1 000 x 16 MiB rayon loop of test_convert_by_padding_empty_byte runs 4 x 1 MiB per thread)
    primitives/tests/helpers_test.rs    (the 9 calculate_roots_of_unity tests -- the roots are generated on the GPU -- and the 5 G1 curve /
                                         validation tests; the other 19 -- pad_payload, is_zeroed, the G2 checks, compute_challenge ... --
                                         are in tests/test_reference_helpers.py, the fixtures also in tests/test_oracle.py)
    verifier/tests/tests.rs             (the 4 tests not already in tests/test_gpu_verifier.py: identity points, intermediate point
                                         validation, zero commitment, random inputs)

The reference's tests use the mainnet SRS file (absent from the reference tree, .MISSING_LARGE_BLOBS) and consts::G2_TAU; here the SRS
is the known-tau one generated on the GPU and [tau]G2 is passed explicitly -- the tests assert algebraic properties (equivalence of
commitment forms, f(w^i) = f_i, error values), which hold for any setup.  Random sizes are drawn from a fixed seed inside the
reference's ranges."""
import hashlib
import random
import threading

import numpy as np
import pytest

import pyref
from pyref import R_

pytestmark = pytest.mark.gpu

TAU = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % R_
SRS_POINTS = 131072                                                   # kzg_test.rs:11-16 loads 131 072 points
MAINNET_SRS_G1_SIZE = 268435456                                       # primitives/src/consts.rs:66
GETTYSBURG_PREFIX = b"Fourscore and seven years ago our fathers brought forth, on this continent, a new nation, conceived in liberty"


@pytest.fixture(scope="module")
def k():
    import rust_kzg_bn254_amd as k
    k.load()
    k.default_context()
    return k


@pytest.fixture(scope="module")
def srs(k):
    """SRS_INSTANCE (kzg_test.rs:11-16: 131 072 mainnet points) -- here 131 072 known-tau points."""
    return k.SRS.generate(TAU, SRS_POINTS)


@pytest.fixture(scope="module")
def g2_tau(k):
    return k.helpers.g2_mul_generator(k.fr.fr_from_int(TAU))


def fr(k, v):
    return k.fr.fr_from_int(v)


def g1(v):
    return np.array(pyref.point_to_wire(pyref.ec_mul(v % R_, (1, 2)) if v % R_ else None), dtype=np.uint64)


# ---- prover/tests/kzg_test.rs -----------------------------------------------------------------------------------------------------
def test_srs_setup_errors(k, golden_dir):
    """kzg_test.rs:19-28"""
    import os
    with pytest.raises(k.errors.GenericError, match="Number of points to load exceeds SRS order."):
        k.SRS.new(os.path.join(golden_dir, "g1.point"), 3000, 3001)


def test_evaluate_polynomial_in_evaluation_form_random_blob_all_indexes(k):
    """kzg_test.rs:31-55: f(w^i) == f_i at EVERY index of a random blob (length in 35 .. 40 000 bytes)."""
    rng = random.Random(31)
    kzg = k.KZG.new()
    blob_length = rng.randrange(35, 40000)
    blob = k.Blob.from_raw_data(bytes(rng.randrange(32, 127) for _ in range(blob_length)))
    poly = blob.to_polynomial_eval_form()
    kzg.calculate_and_store_roots_of_unity(len(blob))
    roots = kzg.get_roots_of_unities()
    evals = poly.evaluations()
    # every index through the batched evaluation entry (one launch), and a sample through the single call the reference makes
    n = poly.len_underlying_blob_field_elements()
    ys = k.helpers.evaluate_blobs_in_evaluation_form_batch([blob] * n, [roots[i] for i in range(n)])
    assert np.array_equal(np.asarray(ys), evals[:n])
    for i in list(range(0, n, max(1, n // 25))) + [n - 1]:
        z = kzg.get_nth_root_of_unity(i)
        assert np.array_equal(k.helpers.evaluate_polynomial_in_evaluation_form(poly, z), evals[i]), i


def test_commit_coeff_form_and_eval_form_equivalence(k, srs):
    """kzg_test.rs:57-89: the SAME polynomial committed in coefficient form and in evaluation form (random blob of 50 .. 500 000 bytes)."""
    rng = random.Random(57)
    blob = k.Blob.from_raw_data(rng.randbytes(rng.randrange(50, 500000)))
    poly_coeff = blob.to_polynomial_coeff_form()
    poly_eval_from_coeff = poly_coeff.to_eval_form()
    kzg = k.KZG.new()
    kzg.calculate_and_store_roots_of_unity(len(blob))
    commitment_coeff = kzg.commit_coeff_form(poly_coeff, srs)
    commitment_eval = kzg.commit_eval_form(poly_eval_from_coeff, srs)
    assert np.array_equal(commitment_coeff, commitment_eval)
    assert not k.fr.g1_is_identity(commitment_coeff)


def test_calculate_and_store_roots_of_unity(k):
    """kzg_test.rs:91-124"""
    kzg = k.KZG.new()
    assert len(kzg.get_roots_of_unities()) == 0
    for blob_length in (32, 50000, MAINNET_SRS_G1_SIZE):
        kzg.calculate_and_store_roots_of_unity(blob_length)
        assert len(kzg.get_roots_of_unities()) > 0


def test_g1_ifft_non_power_of_two_error(k, srs):
    """kzg_test.rs:126-158"""
    with pytest.raises(k.errors.FFTError, match="length provided is not a power of 2"):
        k.KZG.new().g1_ifft(15, srs)


def test_compute_blob_proof_invalid_commitment(k, srs):
    """kzg_test.rs:160-197: a commitment that is not on the curve ((1, 1)) is rejected."""
    kzg = k.KZG.new()
    blob = k.Blob.from_raw_data(b"test data for invalid commitment")
    kzg.calculate_and_store_roots_of_unity(len(blob))
    invalid = np.array(pyref.point_to_wire((1, 1)), dtype=np.uint64)
    assert not k.helpers.is_on_curve_g1(invalid)
    with pytest.raises(k.errors.KzgError):
        kzg.compute_blob_proof(blob, invalid, srs)


# ---- primitives/tests/polynomial_test.rs ------------------------------------------------------------------------------------------
RAW_32 = bytes([42, 212, 238, 227, 192, 237, 178, 128, 19, 108, 50, 204, 87, 81, 63, 120, 232, 27, 116, 108, 74, 168, 109, 84, 89, 9, 6, 233, 144, 200, 125, 40])


def test_to_fr_array(k, gettysburg):
    """polynomial_test.rs:13-47"""
    raw = RAW_32 + bytes(30)
    blob = k.Blob.from_raw_data(raw)
    poly = blob.to_polynomial_coeff_form()
    assert poly.to_bytes_be()[:len(blob.data())] == blob.data()
    assert blob.to_raw_data() == raw
    long_blob = k.Blob.from_raw_data(gettysburg)
    assert long_blob.to_polynomial_coeff_form().to_bytes_be()[:len(long_blob.data())] == long_blob.data()


def test_transform_form(k):
    """polynomial_test.rs:49-66: coefficient form -> evaluation form (GPU NTT) -> coefficient form (GPU INTT) returns the bytes."""
    blob = k.Blob.from_raw_data(RAW_32)
    poly_coeff = blob.to_polynomial_coeff_form()
    poly_coeff_back = poly_coeff.to_eval_form().to_coeff_form()
    assert poly_coeff_back.to_bytes_be()[:len(blob.data())] == blob.data()


def test_polynomial_lengths(k):
    """polynomial_test.rs:68-84: padded to the next power of two."""
    three = np.stack([fr(k, 1), fr(k, 2), fr(k, 3)])
    assert len(k.PolynomialCoeffForm(three).coeffs()) == 4
    assert len(k.PolynomialEvalForm(three).evaluations()) == 4


def test_transform_length_stays_same(k):
    """polynomial_test.rs:86-97"""
    poly_coeff = k.PolynomialCoeffForm(np.stack([fr(k, 1), fr(k, 2), fr(k, 3)]))
    poly_coeff_back = poly_coeff.to_eval_form().to_coeff_form()
    assert len(poly_coeff.coeffs()) == len(poly_coeff_back.coeffs())
    assert np.array_equal(poly_coeff.coeffs(), poly_coeff_back.coeffs())


def test_transform_form_large_blob(k, gettysburg):
    """polynomial_test.rs:99-111"""
    blob = k.Blob.from_raw_data(gettysburg)
    poly_coeff_back = blob.to_polynomial_coeff_form().to_eval_form().to_coeff_form()
    assert poly_coeff_back.to_bytes_be()[:len(blob.data())] == blob.data()


# ---- primitives/tests/blob_test.rs ------------------------------------------------------------------------------------------------
def test_is_empty(k):
    """blob_test.rs:14-21"""
    assert k.Blob.from_raw_data(b"").is_empty()
    assert not k.Blob.from_raw_data(b"hi").is_empty()


def test_validate_blob_data_as_canonical_field_elements(k, gettysburg):
    """blob_test.rs:23-59 (Blob::new validates length and canonicity; pad_payload makes any bytes valid)."""
    pad = k.helpers.pad_payload
    k.Blob(pad(gettysburg[0:62]))
    with pytest.raises(k.errors.KzgError):
        k.Blob(gettysburg[0:64])                                       # not valid elements
    with pytest.raises(k.errors.KzgError):
        k.Blob(gettysburg[0:3])                                        # not a multiple of 32
    test_3 = bytes([0xFF] * 32)
    with pytest.raises(k.errors.KzgError):
        k.Blob(test_3)
    assert len(pad(test_3)) % 32 == 0
    k.Blob(pad(test_3))
    test_4 = bytes([0xFF] * 62)
    with pytest.raises(k.errors.KzgError):
        k.Blob(test_4)
    k.Blob(pad(test_4))
    random_blob = pad(random.Random(23).randbytes(16252928))
    assert len(random_blob) == 16 * 1024 * 1024
    k.Blob(random_blob)


def test_from_padded_bytes_unchecked(k, gettysburg):
    """blob_test.rs:61-70"""
    blob = k.Blob.from_raw_data(gettysburg[0:31])
    blob_unchecked = k.Blob(k.helpers.pad_payload(gettysburg[0:31]))
    assert blob == blob_unchecked
    assert blob == k.Blob.from_padded_unchecked(k.helpers.pad_payload(gettysburg[0:31]))


def test_convert_by_padding_empty_byte(k, gettysburg):
    """blob_test.rs:72-101; the reference runs its random loop on the rayon pool: here four threads share the module."""
    blob = k.Blob.from_raw_data(b"hi")
    assert blob.data() == bytes([0, 104, 105] + [0] * 29)
    assert blob.data() == k.helpers.pad_payload(b"hi")
    blob = k.Blob.from_raw_data(gettysburg)
    assert len(blob.to_raw_data()) == 1488
    errors = []

    def work(seed):
        rng = random.Random(seed)
        for _ in range(4):
            raw = rng.randbytes(rng.randrange(1, 1 << 20))
            b = k.Blob.from_raw_data(raw)
            back = b.to_raw_data()
            if not (len(raw) > len(back) - 32 and len(raw) <= len(back)) or back[:len(raw)] != raw:
                errors.append(seed)
    threads = [threading.Thread(target=work, args=(s,)) for s in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors


# ---- verifier/tests/tests.rs (the tests tests/test_gpu_verifier.py does not already mirror) ----------------------------------------
def test_individual_verify_proof_with_identity_points(k, g2_tau):
    """tests.rs:383-406: identity commitment / proof are accepted as INPUTS (the result is a bool, not an error)."""
    rng = random.Random(383)
    identity = np.zeros(8, dtype=np.uint64)
    valid_proof, valid_commitment = g1(rng.randrange(1, R_)), g1(rng.randrange(1, R_))
    one = fr(k, 1)
    for c, p in ((identity, valid_proof), (valid_commitment, identity), (identity, identity)):
        assert k.verify_proof(c, p, one, one, g2_tau) in (True, False)
        assert k.verify_proof(c, p, one, one) in (True, False)              # consts::G2_TAU, as the reference calls it


def test_verify_proof_intermediate_point_validation(k, g2_tau):
    """tests.rs:408-447: a constant polynomial (commitment = value G1, proof = identity) verifies at any z, under ANY setup."""
    value_fr, z_fr = fr(k, 42), fr(k, 13)
    commitment = g1(42)
    identity = np.zeros(8, dtype=np.uint64)
    assert k.verify_proof(commitment, identity, value_fr, z_fr) is True
    assert k.verify_proof(commitment, identity, value_fr, z_fr, g2_tau) is True
    assert k.verify_proof(commitment, g1(100), value_fr, z_fr) is False
    assert k.verify_proof(commitment, g1(100), value_fr, z_fr, g2_tau) is False
    assert k.verify_proof(commitment, g1(200), fr(k, 999), z_fr) in (True, False)      # no "commitment-value relationship" error exists


def test_verify_proof_zero_commitment_edge_case(k, g2_tau):
    """tests.rs:449-478"""
    rng = random.Random(449)
    identity = np.zeros(8, dtype=np.uint64)
    assert k.verify_proof(identity, g1(rng.randrange(1, R_)), fr(k, 0), fr(k, 1)) in (True, False)
    assert k.verify_proof(identity, identity, fr(k, 0), fr(k, 1), g2_tau) is True       # 0 = 0 * (tau - z)


def test_verify_proof_edge_cases_with_valid_inputs(k, g2_tau):
    """tests.rs:480-507: 100 random (commitment, proof, value, z): never an error, and -- a random proof being wrong -- never accepted."""
    rng = random.Random(480)
    for _ in range(100):
        c, p = g1(rng.randrange(1, R_)), g1(rng.randrange(1, R_))
        assert k.verify_proof(c, p, fr(k, rng.randrange(R_)), fr(k, rng.randrange(R_)), g2_tau) is False
    # and the positive control the reference's loop lacks: p(X) = a + b X, proof = [b] G1, value = a + b z
    a, b, z = rng.randrange(R_), rng.randrange(R_), rng.randrange(R_)
    assert k.verify_proof(g1(a + b * TAU), g1(b), fr(k, (a + b * z) % R_), fr(k, z), g2_tau) is True


# ---- primitives/tests/helpers_test.rs: calculate_roots_of_unity (generated on the GPU) and the G1 point checks -------------------------
def _ints(roots):
    return pyref.frs_from_mont(np.asarray(roots))


def test_calculate_roots_of_unity_error_zero_length(k):
    """helpers_test.rs:28-42"""
    with pytest.raises(k.errors.GenericError, match="Length of data after padding is 0"):
        k.helpers.calculate_roots_of_unity(0)


def test_calculate_roots_of_unity_error_oversized_input(k):
    """helpers_test.rs:44-60"""
    with pytest.raises(k.errors.GenericError, match="the length of data after padding is not valid with respect to the SRS"):
        k.helpers.calculate_roots_of_unity((MAINNET_SRS_G1_SIZE + 1) * 32)


def test_calculate_roots_of_unity_basic_functionality(k):
    """helpers_test.rs:62-89"""
    r32 = k.helpers.calculate_roots_of_unity(32)
    assert len(r32) == 1 and _ints(r32) == [1]
    assert len(k.helpers.calculate_roots_of_unity(64)) == 2
    r96 = k.helpers.calculate_roots_of_unity(96)
    assert len(r96) == 4 and _ints(r96)[0] == 1


def test_calculate_roots_of_unity_mathematical_properties(k):
    """helpers_test.rs:91-130"""
    for length, n in ((3 * 32, 4), (5 * 32, 8)):
        roots = _ints(k.helpers.calculate_roots_of_unity(length))
        assert len(roots) == n and roots[0] == 1
        assert all(pow(w, n, R_) == 1 for w in roots)


def test_calculate_roots_of_unity_powers_of_two(k):
    """helpers_test.rs:132-178"""
    for count in (1, 2, 4, 8, 16, 32):
        roots = _ints(k.helpers.calculate_roots_of_unity(count * 32))
        assert len(roots) == count and roots[0] == 1 and len(set(roots)) == count


def test_calculate_roots_of_unity_boundary_conditions(k):
    """helpers_test.rs:180-222 (the largest accepted length, 2^28 elements = 8 GiB of roots, is only checked for its status here: the
    reference's own test accepts either outcome there)"""
    assert len(k.helpers.calculate_roots_of_unity(1)) == 1
    assert len(k.helpers.calculate_roots_of_unity(32)) == 1
    n_out = __import__("ctypes").c_size_t(0)
    rc = k._lib.load().kzg_calculate_roots_of_unity(k.default_context().handle, MAINNET_SRS_G1_SIZE * 32, None, 0, __import__("ctypes").byref(n_out))
    assert n_out.value == MAINNET_SRS_G1_SIZE and rc == k._lib.ERR_INVALID_ARG          # size query: accepted length, no buffer given


def test_calculate_roots_of_unity_consistency(k):
    """helpers_test.rs:224-243"""
    a, b, c = (k.helpers.calculate_roots_of_unity(100) for _ in range(3))
    assert np.array_equal(a, b) and np.array_equal(b, c)


def test_calculate_roots_of_unity_large_valid_inputs(k):
    """helpers_test.rs:245-283: w^n == 1 for EVERY root at 1 kB .. 1 MB of data."""
    for length in (1000, 10000, 100000, 1000000):
        roots = _ints(k.helpers.calculate_roots_of_unity(length))
        n = len(roots)
        assert n > 0 and roots[0] == 1
        assert all(pow(w, n, R_) == 1 for w in roots)
        assert roots[1] == pyref.root_of_unity(n.bit_length() - 1) and all(roots[i + 1] == roots[i] * roots[1] % R_ for i in range(0, n - 1, 97))


def test_calculate_roots_of_unity_specific_error_conditions(k):
    """helpers_test.rs:285-322: a large valid length (2^23 elements here: 256 MiB of roots) succeeds."""
    roots = k.helpers.calculate_roots_of_unity((1 << 23) * 32)
    assert len(roots) == 1 << 23 and pyref.fr_from_mont(roots[0]) == 1
    w = pyref.fr_from_mont(roots[1])
    assert pow(w, 1 << 23, R_) == 1 and pow(w, 1 << 22, R_) == R_ - 1
    assert pyref.fr_from_mont(roots[(1 << 23) - 1]) == pow(w, (1 << 23) - 1, R_)


def test_g1_is_on_curve(k):
    """helpers_test.rs:324-336"""
    rng = random.Random(324)
    for _ in range(200):
        x, y = pyref.ec_mul(rng.randrange(1, R_), (1, 2))
        assert k.helpers.is_on_curve_g1(np.array(pyref.point_to_wire((x, y)), dtype=np.uint64))
        assert not k.helpers.is_on_curve_g1(np.array(pyref.point_to_wire(((x + 1) % pyref.P, y)), dtype=np.uint64))


def test_validate_g1_point_valid_point(k):
    """helpers_test.rs:622-634"""
    rng = random.Random(622)
    for _ in range(10):
        k.helpers.validate_g1_point(g1(rng.randrange(1, R_)))


def test_validate_g1_point_identity_point(k):
    """helpers_test.rs:636-641"""
    k.helpers.validate_g1_point(np.zeros(8, dtype=np.uint64))


def test_validate_g1_point_invalid_curve_point(k):
    """helpers_test.rs:643-672"""
    with pytest.raises(k.errors.NotOnCurveError, match="G1 point not on curve"):
        k.helpers.validate_g1_point(np.array(pyref.point_to_wire((1, 1)), dtype=np.uint64))


def test_validate_g1_point_generator(k):
    """helpers_test.rs:674-679"""
    k.helpers.validate_g1_point(np.array(pyref.point_to_wire((1, 2)), dtype=np.uint64))
