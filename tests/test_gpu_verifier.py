"""-m gpu: the verifier row end to end, mirroring verifier/tests/tests.rs.  Commitments and proofs come from the HIP
prover path (MSM / NTT / proof pipeline on the GPU), verification = GPU barycentric evaluation + batched GPU MSMs + the
library's O(1) host pairing.  The reference's tests use the mainnet SRS file and consts::G2_TAU; that file is not part of
the reference tree available here, so the setup is SRS.generate(TAU) on the GPU with [TAU]G2 passed explicitly.
Independent of the oracle: a pairing check is a cryptographic proof that commitment and proof are correct."""
import random

import numpy as np
import pytest

import pyref
from pyref import R_

pytestmark = pytest.mark.gpu

TAU = int.from_bytes(__import__("hashlib").sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % R_


@pytest.fixture(scope="module")
def k():
    import rust_kzg_bn254_amd as k
    k.load()
    k.default_context()
    return k


@pytest.fixture(scope="module")
def srs(k):
    return k.SRS.generate(TAU, 4096)


@pytest.fixture(scope="module")
def g2_tau(k):
    return k.helpers.g2_mul_generator(k.fr.fr_from_int(TAU))


def _prove(k, srs, raw):
    kz = k.KZG.new()
    blob = k.Blob.from_raw_data(raw)
    kz.calculate_and_store_roots_of_unity(len(blob))
    poly = blob.to_polynomial_eval_form()
    commitment = kz.commit_eval_form(poly, srs)
    proof = kz.compute_blob_proof(blob, commitment, srs)
    return blob, commitment, proof


def test_compute_kzg_proof_and_verify(k, srs, g2_tau, gettysburg):
    """tests.rs:28-77: every 6th index of the Gettysburg polynomial (each verification costs two host pairings)."""
    kz = k.KZG.new()
    blob = k.Blob.from_raw_data(gettysburg)
    poly = blob.to_polynomial_eval_form()
    kz.calculate_and_store_roots_of_unity(len(blob))
    commitment = kz.commit_eval_form(poly, srs)
    m = poly.len_underlying_blob_field_elements()
    for index in range(0, len(poly) - 1, 6):
        proof = kz.compute_proof_with_known_z_fr_index(poly, index, srs)
        value, z = poly.get_evalualtion(index), kz.get_nth_root_of_unity(index)
        assert k.verify_proof(commitment, proof, value, z, g2_tau) is True
        other = kz.get_nth_root_of_unity((index + 1) % m)
        assert k.verify_proof(commitment, proof, value, other, g2_tau) is False
    assert k.verify_proof(commitment, proof, value, z) is False        # mainnet G2_TAU is a different setup


def test_random_blobs_single_and_batch(k, srs, g2_tau):
    """tests.rs:80-132 and :135-192 (12 blobs instead of 100; lengths 35..50000 bytes)."""
    rnd = random.Random(5)
    blobs, commitments, proofs = [], [], []
    for _ in range(12):
        raw = bytes(rnd.randrange(32, 127) for _ in range(rnd.randrange(35, 50000)))
        b, c, p = _prove(k, srs, raw)
        blobs.append(b); commitments.append(c); proofs.append(p)
    for b, c, p in list(zip(blobs, commitments, proofs))[:3]:
        assert k.verify_blob_kzg_proof(b, c, p, g2_tau) is True
    assert k.verify_blob_kzg_proof(blobs[0], commitments[0], proofs[1], g2_tau) is False
    assert k.verify_blob_kzg_proof_batch(blobs, commitments, proofs, g2_tau) is True
    bad_blobs = blobs[:-1] + [k.Blob.from_raw_data(b"random")]
    assert k.verify_blob_kzg_proof_batch(bad_blobs, commitments, proofs, g2_tau) is False
    rand_pt = lambda s: np.array(pyref.point_to_wire(pyref.ec_mul(s, (1, 2))), dtype=np.uint64)   # noqa: E731
    bad_commitments = commitments[:-1] + [rand_pt(123457)]
    assert k.verify_blob_kzg_proof_batch(blobs, bad_commitments, proofs, g2_tau) is False
    bad_proofs = proofs[:-1] + [rand_pt(7654321)]
    assert k.verify_blob_kzg_proof_batch(blobs, commitments, bad_proofs, g2_tau) is False
    assert k.verify_blob_kzg_proof_batch(bad_blobs, bad_commitments, bad_proofs, g2_tau) is False
    assert k.verify_blob_kzg_proof_batch(blobs, commitments, proofs) is False          # wrong setup


def test_compute_multiple_kzg_proof(k, srs, g2_tau, gettysburg):
    """tests.rs:195-237."""
    b1, c1, p1 = _prove(k, srs, gettysburg)
    b2, c2, p2 = _prove(k, srs, b"17704588942648532530972307366230787358793284390049200127770755029903181125533")
    assert k.verify_blob_kzg_proof_batch([b1, b2], [c1, c2], [p1, p2], g2_tau) is True
    assert k.verify_blob_kzg_proof_batch([b1, b2], [c2, c1], [p1, p2], g2_tau) is False
    assert k.verify_blob_kzg_proof_batch([], [], [], g2_tau) is True


def test_kzg_zero_blob(k, srs, g2_tau):
    """tests.rs:240-269: the all-zero blob commits to the identity and verifies, singly and in a batch."""
    blob, commitment, proof = _prove(k, srs, bytes(62))
    assert blob.data() == bytes(64)
    assert k.fr.g1_is_identity(commitment)
    assert k.verify_blob_kzg_proof_batch([blob], [commitment], [proof], g2_tau) is True
    assert k.verify_blob_kzg_proof(blob, commitment, proof, g2_tau) is True


def test_kzg_batch_proof_with_infinity(k, srs, g2_tau, gettysburg):
    """tests.rs:272-311: identity proofs are valid inputs (result is a bool, not an error)."""
    b1, c1, p1 = _prove(k, srs, gettysburg)
    ident = np.zeros(8, dtype=np.uint64)
    assert k.verify_blob_kzg_proof_batch([b1], [c1], [ident], g2_tau) is False
    b2, c2, _ = _prove(k, srs, b"second input")
    assert k.verify_blob_kzg_proof_batch([b1, b2], [c1, c2], [p1, ident], g2_tau) in (True, False)


def test_kzg_batch_proof_invalid_curve_points(k, srs, g2_tau, gettysburg):
    """tests.rs:314-380: off-curve commitments / proofs are rejected with NotOnCurveError; length mismatch with GenericError."""
    b, c, p = _prove(k, srs, gettysburg)
    off = np.array(pyref.point_to_wire((1, 3)), dtype=np.uint64)
    p_plus_1 = p.copy()
    x, y = pyref.point_from_wire(p)
    p_plus_1[4:] = np.array(pyref.point_to_wire((x, (y + 1) % pyref.P)), dtype=np.uint64)[4:]
    cases = [([off, c], [p, p]), ([c, c], [off, p]), ([off, c], [off, p]), ([c, off], [p, off]), ([c, off], [p, p_plus_1]),
             ([off, off], [off, p_plus_1]), ([c, c], [p, p_plus_1])]
    for commitments, proofs in cases:
        with pytest.raises(k.errors.NotOnCurveError):
            k.verify_blob_kzg_proof_batch([b, b], commitments, proofs, g2_tau)
    with pytest.raises(k.errors.NotOnCurveError):
        k.verify_blob_kzg_proof(b, off, p, g2_tau)
    with pytest.raises(k.errors.NotOnCurveError):
        k.KZG.new().compute_blob_proof(b, off, srs)
    with pytest.raises(k.errors.GenericError, match="length's of the input are not the same"):
        k.verify_blob_kzg_proof_batch([b, b], [c], [p, p], g2_tau)
    # the C entry point validates too (a binding that skips the host-side loop still gets the reference's error)
    import ctypes as C
    ok = C.c_int32(0)
    one = k.fr.fr_from_int(1).reshape(1, 4)
    rc = k._lib.load().kzg_verify_kzg_proof_batch(k.default_context().handle, k._lib.ptr(off.reshape(1, 8)), k._lib.ptr(one), k._lib.ptr(one),
                                                  k._lib.ptr(p.reshape(1, 8).copy()), k._lib.ptr(one), 1, None, C.byref(ok))
    assert rc == k._lib.ERR_G1_NOT_ON_CURVE


def test_batch_4096_proofs_reuse(k, srs, g2_tau, gettysburg):
    """verify_kzg_proof_batch at n = 4096 (batch.rs:185-256): the same valid (C, z, y, proof) row repeated — the three
    4096-point linear combinations run as one batched GPU MSM; flipping one y makes it fail."""
    kz = k.KZG.new()
    blob = k.Blob.from_raw_data(gettysburg)
    poly = blob.to_polynomial_eval_form()
    kz.calculate_and_store_roots_of_unity(len(blob))
    commitment = kz.commit_eval_form(poly, srs)
    rows = []
    for i in range(4):
        z = k.fr.fr_from_int(1000003 + i)
        proof = kz.compute_proof(poly, z, srs)
        y = k.helpers.evaluate_polynomial_in_evaluation_form(poly, z)
        rows.append((commitment, z, y, proof))
    n = 4096
    sel = [rows[i % 4] for i in range(n)]
    cs, zs, ys, ps = ([r[j] for r in sel] for j in range(4))
    lens = [64] * n
    assert k.verifier.verify_kzg_proof_batch(cs, zs, ys, ps, lens, g2_tau) is True
    ys[2077] = k.fr.fr_from_int(5)
    assert k.verifier.verify_kzg_proof_batch(cs, zs, ys, ps, lens, g2_tau) is False


# ---------------------------------------------------------------------------------------------------------
# verify_blob_kzg_proof_batch behind the C-ABI (BASELINE config 5 end to end)
# ---------------------------------------------------------------------------------------------------------
def _rand_blob(k, rng, n_raw):
    return k.Blob.from_raw_data(rng.integers(32, 127, size=n_raw, dtype=np.uint8).tobytes())


def test_batched_challenges_and_evaluations_match_oracle(k, srs):
    """helpers.rs:613-662 in one call (`kzg_compute_challenges_and_evaluate_polynomial`: host thread pool for the n transcripts, one
    batched GPU launch for the n barycentric evaluations) against the oracle's blob-by-blob restatement: every padded size 2^0 ..
    2^12 (1, 2 and 4 denominators per lane; the 1-element domain), a ragged last chunk, non-canonical chunks (>= r: reduced by
    to_fr_array), an all-zero blob, and a 2^13-element blob (beyond the batched kernel: single-polynomial path)."""
    import oracle as orc
    rng = np.random.default_rng(77)
    blobs = [_rand_blob(k, rng, n_raw) for n_raw in (1, 31, 32, 62, 100, 300, 700, 1500, 3000, 7000, 15000, 31000, 50000, 63000, 126000, 127000)]
    blobs.append(k.Blob.from_padded_unchecked(bytes(range(1, 46))))                      # 45 bytes: ragged second chunk
    blobs.append(k.Blob.from_padded_unchecked(b"\xff" * 96 + bytes(32) + b"\x30" + b"\xee" * 31))   # chunks >= r
    blobs.append(k.Blob.from_padded_unchecked(bytes(4096)))                              # zero polynomial
    blobs.append(_rand_blob(k, rng, 200000))                                             # 8192 padded elements
    sizes = sorted({len(b.to_polynomial_eval_form()) for b in blobs})
    assert sizes[0] == 1 and 4096 in sizes and 8192 in sizes and 2048 in sizes
    cms = [np.array(pyref.point_to_wire(pyref.ec_mul(1000 + i, (1, 2))), dtype=np.uint64) for i in range(len(blobs))]
    cms[3] = np.zeros(8, dtype=np.uint64)                                                # identity commitment
    zs, ys = k.helpers.compute_challenges_and_evaluate_polynomial(blobs, cms)
    rc, zs_want, ys_want = orc.compute_challenges_and_evaluate_polynomial([b.data() for b in blobs], np.stack(cms))
    assert rc == 0
    assert np.array_equal(np.stack(zs), zs_want)
    assert np.array_equal(np.stack(ys), ys_want)
    # the single-blob entry points agree
    zs2, ys2 = k.helpers.compute_challenges_and_evaluate_polynomial_py(blobs[:6], cms[:6])
    assert np.array_equal(np.stack(zs2), zs_want[:6]) and np.array_equal(np.stack(ys2), ys_want[:6])
    # errors (helpers.rs:413, :554-558)
    off = np.array(pyref.point_to_wire((1, 3)), dtype=np.uint64)
    with pytest.raises(k.errors.NotOnCurveError):
        k.helpers.compute_challenges_and_evaluate_polynomial(blobs[:3], [cms[0], off, cms[2]])
    with pytest.raises(k.errors.GenericError, match="Length of data after padding is 0"):
        k.helpers.compute_challenges_and_evaluate_polynomial([blobs[0], k.Blob.from_padded_unchecked(b"")], cms[:2])
    with pytest.raises(k.errors.GenericError, match="length's of the input are not the same or is empty"):
        k.helpers.compute_challenges_and_evaluate_polynomial(blobs[:3], cms[:2])
    assert k.helpers.compute_challenges_and_evaluate_polynomial([], []) == ([], [])


def test_batched_evaluation_with_points_on_the_domain(k):
    """`kzg_evaluate_blobs_in_evaluation_form_batch` with CALLER-given points: z on the domain (helpers.rs:497-504: the evaluation
    itself is returned; the batched kernel flags 1 - z^n == 0 and the blob takes the single-polynomial path), z = 0, z = 1 on a
    1-element domain, random z; all against the oracle."""
    import oracle as orc
    rng = np.random.default_rng(5)
    rnd = random.Random(6)
    blobs, zs = [], []
    for n_raw in (20, 62, 500, 3000, 20000, 63000, 126000):
        b = _rand_blob(k, rng, n_raw)
        n = len(b.to_polynomial_eval_form())
        log_n = n.bit_length() - 1
        w = pyref.root_of_unity(log_n) if log_n else 1
        for z in (pow(w, rnd.randrange(n), R_), pow(w, n - 1, R_), 1, 0, rnd.randrange(R_)):
            blobs.append(b)
            zs.append(k.fr.fr_from_int(z))
    ys = k.helpers.evaluate_blobs_in_evaluation_form_batch(blobs, zs)
    for b, z, y in zip(blobs, zs, ys):
        ev = orc.to_fr_array(b.data())
        n = len(b.to_polynomial_eval_form())
        padded = np.zeros((n, 4), np.uint64); padded[:len(ev)] = ev
        rc, want = orc.evaluate_polynomial_in_evaluation_form(padded, z)
        assert rc == 0 and np.array_equal(y, want), (len(b), pyref.fr_from_mont(z))


def test_verify_blob_kzg_proof_batch_4096_end_to_end(k, srs, g2_tau):
    """verifier/tests/tests.rs:134-192 at BASELINE config 5's size through ONE C-ABI call (`kzg_verify_blob_kzg_proof_batch`):
    4096 random blobs of 35 .. 50000 bytes with their GPU commitments and blob proofs verify; replacing the last blob, the last
    commitment, the last proof, or all three makes the batch fail; a wrong trusted setup fails; the step-by-step form agrees."""
    import oracle as orc
    n = 4096
    rng = np.random.default_rng(4096)
    kz = k.KZG.new()
    blobs, commitments, proofs = [], [], []
    lens = rng.integers(35, 50000, size=n)
    lens[:4] = (35, 49999, 31 * 1024, 31 * 1024 + 1)           # smallest / largest / exactly 1024 elements / one more
    for n_raw in lens:
        blob = _rand_blob(k, rng, int(n_raw))
        kz.calculate_and_store_roots_of_unity(len(blob))
        commitment, proof, _z, _y = kz.commit_and_prove_blob(blob, srs)
        blobs.append(blob); commitments.append(commitment); proofs.append(proof)
    assert k.verify_blob_kzg_proof_batch(blobs, commitments, proofs, g2_tau) is True
    # the transcripts of the batch, against the oracle (first 64 blobs: the oracle evaluates with one inversion per element)
    zs, ys = k.helpers.compute_challenges_and_evaluate_polynomial(blobs[:64], commitments[:64])
    rc, zs_want, ys_want = orc.compute_challenges_and_evaluate_polynomial([b.data() for b in blobs[:64]], np.stack(commitments[:64]))
    assert rc == 0 and np.array_equal(np.stack(zs), zs_want) and np.array_equal(np.stack(ys), ys_want)
    lengths = [len(b.to_polynomial_eval_form()) for b in blobs[:64]]
    assert np.array_equal(k.verifier.compute_r_powers(commitments[:64], zs, ys, proofs[:64], lengths),
                          orc.compute_r_powers(np.stack(commitments[:64]), zs_want, ys_want, np.stack(proofs[:64]), lengths))
    rand_pt = lambda s: np.array(pyref.point_to_wire(pyref.ec_mul(s, (1, 2))), dtype=np.uint64)   # noqa: E731
    bad_blobs = blobs[:-1] + [k.Blob.from_raw_data(b"random")]
    bad_commitments = commitments[:-1] + [rand_pt(123457)]
    bad_proofs = proofs[:-1] + [rand_pt(7654321)]
    assert k.verify_blob_kzg_proof_batch(bad_blobs, commitments, proofs, g2_tau) is False
    assert k.verify_blob_kzg_proof_batch(blobs, bad_commitments, proofs, g2_tau) is False
    assert k.verify_blob_kzg_proof_batch(blobs, commitments, bad_proofs, g2_tau) is False
    assert k.verify_blob_kzg_proof_batch(bad_blobs, bad_commitments, bad_proofs, g2_tau) is False
    assert k.verify_blob_kzg_proof_batch(blobs, commitments, proofs) is False                    # mainnet G2_TAU: another setup
    assert k.verifier.verify_blob_kzg_proof_batch_py(blobs[:200], commitments[:200], proofs[:200], g2_tau) is True
    assert k.verifier.verify_blob_kzg_proof_batch_py(blobs[:200], commitments[:199] + [rand_pt(5)], proofs[:200], g2_tau) is False
    # a corrupted blob in the MIDDLE of the batch (one flipped byte)
    data = bytearray(blobs[2000].data()); data[40] ^= 1
    mid = blobs[:2000] + [k.Blob.from_padded_unchecked(bytes(data))] + blobs[2001:]
    assert k.verify_blob_kzg_proof_batch(mid, commitments, proofs, g2_tau) is False


def test_batch_front_end_multi_round_and_multi_chunk_paths(k, srs, g2_tau):
    """A batch larger than one GPU round (256 MiB of packed blobs) is processed in rounds, a round in chunks of 16 MiB whose uploads
    run beside the hashing: with the test hooks KZG_VB_GROUP_BYTES / KZG_VB_CHUNK_BYTES a 60-blob batch takes both paths (several rounds,
    several chunks per round, a 2^13-element blob in the middle of a round) in a fresh process; results must equal the one-round run."""
    import subprocess, sys, os, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent('''
        import sys, hashlib
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import numpy as np, torch
        import rust_kzg_bn254_amd as k, pyref
        k.load(); k.default_context()
        TAU = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") %% pyref.R_
        srs = k.SRS.generate(TAU, 4096)
        g2 = k.helpers.g2_mul_generator(k.fr.fr_from_int(TAU))
        rng = np.random.default_rng(99)
        kz = k.KZG.new(); blobs = []; cs = []; ps = []
        for i in range(60):
            n_raw = int(rng.integers(35, 60000))
            b = k.Blob.from_raw_data(rng.integers(32, 127, size=n_raw, dtype=np.uint8).tobytes())
            kz.calculate_and_store_roots_of_unity(len(b))
            c, p, _, _ = kz.commit_and_prove_blob(b, srs)
            blobs.append(b); cs.append(c); ps.append(p)
        big = k.Blob.from_raw_data(rng.integers(32, 127, size=200000, dtype=np.uint8).tobytes())     # 8192 elements: single-polynomial path
        zs, ys = k.helpers.compute_challenges_and_evaluate_polynomial(blobs[:30] + [big] + blobs[30:], cs[:30] + [cs[0]] + cs[30:])
        print("ZY", hashlib.sha256(np.stack(zs).tobytes() + np.stack(ys).tobytes()).hexdigest())
        print("OK", k.verify_blob_kzg_proof_batch(blobs, cs, ps, g2), k.verify_blob_kzg_proof_batch(blobs, cs[1:] + cs[:1], ps, g2))
    ''') % (root, os.path.join(root, "tests"))
    outs = []
    for env_extra in ({}, {"KZG_VB_GROUP_BYTES": str(300000), "KZG_VB_CHUNK_BYTES": str(70000)}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith(("ZY", "OK"))])
    assert outs[0] == outs[1] and outs[0][1] == "OK True False", outs
