"""-m gpu: the C++ host mirror of the reference API (include/kzg_bn254_mi355x.hpp: KZG, SRS, Blob, PolynomialEvalForm / CoeffForm, KzgError,
verify_proof, verify_blob_kzg_proof, verify_blob_kzg_proof_batch -- the reference's names, arguments and error texts over the C-ABI).

tests/cpp/reference_tests.cpp restates the reference's own live tests (prover/tests/kzg_test.rs, verifier/tests/tests.rs) against it; this
file builds it with g++, runs it as its own process (no Python, no torch on the product side) and cross-checks the values it prints:
the commitment of the Gettysburg blob against sum_i c_i tau^i G1 and its proofs against ((f(tau) - y) / (tau - z)) G1, both by big-integer
arithmetic (tests/pyref.py), and all four values against the Python mirror."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import pyref
from pyref import R_

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAU = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % R_
TESTS = ["test_srs_setup_errors", "test_evaluate_polynomial_in_evaluation_form_random_blob_all_indexes",
         "test_commit_coeff_form_and_eval_form_equivalence", "test_calculate_and_store_roots_of_unity", "test_g1_ifft_non_power_of_two_error",
         "test_compute_blob_proof_invalid_commitment", "test_compute_kzg_proof", "test_compute_kzg_proof_random_100_blobs", "test_kzg_zero_blob",
         "test_multiple_proof_random_100_blobs", "test_kzg_batch_proof_invalid_curve_points", "test_blob_padding_and_validation",
         # primitives/tests/helpers_test.rs through the C++ mirror's helpers namespace
         "test_g2_is_on_curve", "test_get_num_element", "test_pad_payload", "test_to_fr_array", "test_is_zeroed", "test_primitive_roots_of_unity",
         "test_validate_g1_point_and_g2_point", "test_compute_challenge_comprehensive", "test_compute_challenges_and_evaluate_polynomial",
         "test_read_g1_point_and_lincomb", "test_compute_quotient_eval_on_domain"]


def build(tmp_path):
    exe = str(tmp_path / "reference_tests")
    libdir = os.path.join(ROOT, "rust-kzg-bn254_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "reference_tests.cpp"), "-L" + libdir, "-lkzg_bn254_mi355x", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def wire_hex(a):
    return "".join("%016x" % int(w) for w in np.asarray(a, dtype=np.uint64).reshape(-1))


def test_reference_tests_in_cpp_and_their_values(tmp_path):
    exe = build(tmp_path)
    res = subprocess.run([exe, os.path.join(ROOT, "tests", "golden"), "%064x" % TAU], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.returncode, res.stdout[-3000:], res.stderr[-1500:])
    lines = res.stdout.splitlines()
    for name in TESTS:
        assert "ok     " + name in lines, (name, res.stdout[-3000:])
    assert "0 failure(s)" in lines
    vals = {ln.split()[1]: ln.split()[2] for ln in lines if ln.startswith("value ")}

    # big-integer expectations: the blob's 64 evaluations -> coefficients (O(n^2) DFT over the integers mod r) -> f(tau)
    raw = open(os.path.join(ROOT, "tests", "golden", "gettysburg.txt"), "rb").read()
    evals = pyref.to_fr_array(pyref.pad_payload(raw))
    n = pyref.next_pow2(len(evals))
    evals = evals + [0] * (n - len(evals))
    coeffs = pyref.dft(evals, inverse=True)
    f_tau = pyref.poly_eval(coeffs, TAU)
    g1 = (1, 2)
    assert vals["commitment"] == wire_hex(pyref.point_to_wire(pyref.ec_mul(f_tau, g1)))
    w = pyref.root_of_unity(n.bit_length() - 1)
    z7 = pow(w, 7, R_)
    q7 = (f_tau - evals[7]) * pow(TAU - z7, -1, R_) % R_
    assert vals["proof_index_7"] == wire_hex(pyref.point_to_wire(pyref.ec_mul(q7, g1)))
    # KZG::compute_quotient_eval_on_domain (kzg.rs:237-260) at z = w^7, value = f_7: the literal sum over the other roots
    q_dom = sum((evals[i] - evals[7]) * pow(w, i, R_) * pow((z7 - pow(w, i, R_)) * z7, -1, R_) for i in range(n) if i != 7) % R_
    assert vals["quotient_eval_on_domain"] == wire_hex(pyref.fr_to_mont(q_dom))
    z = pyref.fr_from_mont(np.array([int(vals["challenge"][16 * i:16 * i + 16], 16) for i in range(4)], dtype=np.uint64))
    y = pyref.poly_eval(coeffs, z)
    qz = (f_tau - y) * pow(TAU - z, -1, R_) % R_
    assert vals["blob_proof"] == wire_hex(pyref.point_to_wire(pyref.ec_mul(qz, g1)))

    # and the Python mirror of the same API (its challenge is checked against hashlib in tests/test_challenge_host.py)
    import rust_kzg_bn254_amd as k
    srs = k.SRS.generate(TAU, 4096)
    try:
        kzg = k.KZG.new()
        blob = k.Blob.from_raw_data(raw)
        kzg.calculate_and_store_roots_of_unity(len(blob))
        c = kzg.commit_blob(blob, srs)
        assert vals["commitment"] == wire_hex(c)
        assert vals["proof_index_7"] == wire_hex(kzg.compute_proof_with_known_z_fr_index(blob.to_polynomial_eval_form(), 7, srs))
        proof, zz, _ = kzg.compute_blob_proof(blob, c, srs, want_zy=True)
        assert vals["blob_proof"] == wire_hex(proof) and vals["challenge"] == wire_hex(zz)
    finally:
        srs.close()
