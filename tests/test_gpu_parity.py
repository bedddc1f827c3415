"""-m gpu: parity of the HIP path (through the C-ABI) against the oracle, the reference's golden vectors and
size-independent properties.  Bit-exact (integer arithmetic): np.array_equal on the wire limbs."""
import ctypes as C
import os
import random

import numpy as np
import pytest

import oracle as orc
import pyref
from pyref import P, R_

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAU = int.from_bytes(__import__("hashlib").sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % R_


@pytest.fixture(scope="module")
def k():
    import rust_kzg_bn254_amd as k
    k.load()
    k.default_context()          # raises DeviceError if the HIP extension or the GPU is missing
    return k


@pytest.fixture(scope="module")
def ref_srs(k, test_srs_wire):
    return k.SRS(test_srs_wire, order=3000)


def rand_scalars(n, seed, mod=R_):
    rnd = random.Random(seed)
    return pyref.frs_to_mont([rnd.randrange(mod) for _ in range(n)])


def msm_srs(k, srs, scalars, offset=0):
    ctx = srs.ctx
    sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(8, np.uint64); inf = C.c_uint8(7)
    rc = k._lib.load().kzg_msm_g1_srs(ctx.handle, srs.handle, offset, k._lib.ptr(sc), len(sc), k._lib.ptr(out), C.byref(inf))
    assert rc == 0, (rc, ctx.last_error())
    assert inf.value == (0 if out.any() else 1)
    return out


# ---------------------------------------------------------------------------------------------------------
# MSM
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 64, 255, 1000, 3000])
def test_msm_matches_oracle_on_reference_srs(k, ref_srs, test_srs_wire, n):
    sc = rand_scalars(n, n)
    got = msm_srs(k, ref_srs, sc)
    assert np.array_equal(got, orc.msm_pippenger(test_srs_wire[:n], sc))


@pytest.fixture(scope="module")
def small_srs(k, test_srs_wire):
    """1 500 points: below 2^11, so WITHOUT per-bit tables -- its small MSMs run through the sort / buckets / fused first level."""
    return k.SRS(test_srs_wire[:1500], order=1500)


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 64, 255, 1000, 1500])
def test_msm_matches_oracle_through_the_buckets(k, small_srs, test_srs_wire, n):
    sc = rand_scalars(n, 7000 + n)
    assert np.array_equal(msm_srs(k, small_srs, sc), orc.msm_pippenger(test_srs_wire[:n], sc))


@pytest.mark.parametrize("c", [2, 3, 5, 8, 11, 13, 16])
def test_msm_every_window_size(k, ref_srs, test_srs_wire, c):
    n = 700
    sc = rand_scalars(n, 100 + c)
    ref_srs.ctx.set_msm_window(c, 0)
    try:
        got = msm_srs(k, ref_srs, sc)
    finally:
        ref_srs.ctx.set_msm_window(0, 0)
    assert np.array_equal(got, orc.msm_pippenger(test_srs_wire[:n], sc))


def test_msm_offset_and_g1_lincomb(k, ref_srs, test_srs_wire):
    n = 500
    sc = rand_scalars(n, 9)
    want = orc.msm_pippenger(test_srs_wire[1000:1000 + n], sc)
    assert np.array_equal(msm_srs(k, ref_srs, sc, offset=1000), want)
    # helpers::g1_lincomb with caller-provided bases (helpers.rs:328-337)
    assert np.array_equal(k.helpers.g1_lincomb(test_srs_wire[1000:1000 + n], sc), want)
    with pytest.raises(k.errors.MsmError):
        k.helpers.g1_lincomb(test_srs_wire[:10], sc[:9])


def test_msm_edge_cases(k, ref_srs, test_srs_wire, test_srs_points):
    n = 512
    zero = pyref.frs_to_mont([0] * n)
    assert not msm_srs(k, ref_srs, zero).any()                       # zero blob -> identity (verifier tests.rs:239-269)
    ones = pyref.frs_to_mont([1] * n)
    assert np.array_equal(msm_srs(k, ref_srs, ones), orc.msm_pippenger(test_srs_wire[:n], ones))
    rm1 = pyref.frs_to_mont([R_ - 1] * n)
    assert np.array_equal(msm_srs(k, ref_srs, rm1), orc.msm_pippenger(test_srs_wire[:n], rm1))
    onehot = pyref.frs_to_mont([0] * 100 + [12345] + [0] * (n - 101))
    assert np.array_equal(msm_srs(k, ref_srs, onehot), orc.g1_scalar_mul(test_srs_wire[100], pyref.fr_to_mont(12345)))
    assert not msm_srs(k, ref_srs, np.zeros((0, 4), np.uint64)).any()  # empty input
    # duplicate points (tests.rs:343-346): forces P + P in the buckets
    dup = np.repeat(test_srs_wire[7:8], n, axis=0)
    got = k.helpers.g1_lincomb(dup, ones)
    assert pyref.point_from_wire(got) == pyref.ec_mul(n, test_srs_points[7])
    sc = rand_scalars(n, 77)
    assert np.array_equal(k.helpers.g1_lincomb(dup, sc), orc.msm_pippenger(dup, sc))
    # P, -P pairs with equal scalars: forces the identity inside buckets
    pm = np.zeros((n, 8), np.uint64)
    pm[0::2] = test_srs_wire[11]; pm[1::2] = orc.g1_neg(test_srs_wire[11])
    assert not k.helpers.g1_lincomb(pm, pyref.frs_to_mont([5] * n)).any()
    # identity bases (tests.rs:271-311) are skipped
    with_inf = test_srs_wire[:n].copy(); with_inf[::3] = 0
    assert np.array_equal(k.helpers.g1_lincomb(with_inf, sc), orc.msm_pippenger(with_inf, sc))


def test_msm_heavy_bucket_segments(k, ref_srs, test_srs_wire):
    """All scalars equal: every window has a single bucket with n entries -> exercises the segment split."""
    n = 3000
    sc = pyref.frs_to_mont([0x1234567 + (1 << 200)] * n)
    for seg in (0, 7):
        ref_srs.ctx.set_msm_window(0, seg)
        try:
            got = msm_srs(k, ref_srs, sc)
        finally:
            ref_srs.ctx.set_msm_window(0, 0)
        assert np.array_equal(got, orc.msm_pippenger(test_srs_wire[:n], sc))


def test_fold_partials_matches_full_msm(k, ref_srs, test_srs_wire):
    """The multi-GPU shape on one GPU: shard by scalar index, partial XYZZ per shard, host fold."""
    n, G = 2048, 4
    sc = rand_scalars(n, 5)
    lib = k._lib.load(); ctx = ref_srs.ctx
    parts = np.zeros((G, 16), np.uint64)
    for g in range(G):
        lo, hi = g * n // G, (g + 1) * n // G
        s = np.ascontiguousarray(sc[lo:hi])
        rc = lib.kzg_msm_g1_srs_partial(ctx.handle, ref_srs.handle, lo, k._lib.ptr(s), hi - lo, k._lib.ptr(parts[g]))
        assert rc == 0
    out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
    assert lib.kzg_g1_fold_partials(k._lib.ptr(parts), G, k._lib.ptr(out), C.byref(inf)) == 0
    assert np.array_equal(out, orc.msm_pippenger(test_srs_wire[:n], sc))


@pytest.fixture(scope="module")
def tau_srs(k):
    return k.SRS.generate(TAU, 1 << 16)


def test_generated_srs_is_tau_powers(k, tau_srs):
    g1 = tau_srs.g1
    assert pyref.point_from_wire(g1[0]) == (1, 2)
    for i in (1, 2, 63, 64, 65, 4097, 65535):
        assert pyref.point_from_wire(g1[i]) == pyref.ec_mul(pow(TAU, i, R_), (1, 2)), i


@pytest.mark.parametrize("log_n", [12, 16])
def test_msm_known_tau_and_oracle(k, tau_srs, log_n):
    """commit(p) == p(tau) * G1 and bit-equality with the oracle's Pippenger."""
    n = 1 << log_n
    rnd = random.Random(log_n)
    vals = [rnd.randrange(R_) for _ in range(n)]
    sc = pyref.frs_to_mont(vals)
    got = msm_srs(k, tau_srs, sc)
    ptau = sum(v * pow(TAU, i, R_) for i, v in enumerate(vals)) % R_
    assert pyref.point_from_wire(got) == pyref.ec_mul(ptau, (1, 2))
    assert np.array_equal(got, orc.msm_pippenger(tau_srs.g1[:n], sc))


def test_table_mode_and_generic_mode_agree(k, tau_srs):
    """SRS calls use the precomputed window tables; a forced window size switches to the generic path."""
    n = 1 << 14
    sc = rand_scalars(n, 4242)
    a = msm_srs(k, tau_srs, sc)
    tau_srs.ctx.set_msm_window(13, 0)
    try:
        b = msm_srs(k, tau_srs, sc)
    finally:
        tau_srs.ctx.set_msm_window(0, 0)
    assert np.array_equal(a, b)
    assert np.array_equal(a, k.helpers.g1_lincomb(tau_srs.g1[:n], sc))
    # sub-range of the tables
    assert np.array_equal(msm_srs(k, tau_srs, sc[:5000], offset=777), k.helpers.g1_lincomb(tau_srs.g1[777:5777], sc[:5000]))


def test_msm_2_20_known_tau(k):
    """BASELINE config 2 at full size: 2^20 blob-like scalars; property check commit == p(tau) * G1."""
    n = 1 << 20
    srs = k.SRS.generate(TAU, n)
    rng = np.random.default_rng(11)
    raw = rng.integers(32, 127, size=(n, 31), dtype=np.uint8)
    vals = [int.from_bytes(raw[i].tobytes(), "big") for i in range(n)]
    sc = pyref.frs_to_mont(vals)
    got = msm_srs(k, srs, sc)
    ptau, cur = 0, 1
    for v in vals:
        ptau = (ptau + v * cur) % R_
        cur = cur * TAU % R_
    assert pyref.point_from_wire(got) == pyref.ec_mul(ptau, (1, 2))
    # degenerate: all scalars equal to one -> sum of all SRS points = (tau^n - 1)/(tau - 1) * G1
    ones = pyref.frs_to_mont([1]) * np.ones((n, 1), dtype=np.uint64)
    ones = np.ascontiguousarray(np.broadcast_to(pyref.fr_to_mont(1), (n, 4)))
    geo = (pow(TAU, n, R_) - 1) * pow(TAU - 1, -1, R_) % R_
    assert pyref.point_from_wire(msm_srs(k, srs, ones)) == pyref.ec_mul(geo, (1, 2))
    srs.close()


# ---------------------------------------------------------------------------------------------------------
# NTT
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 6, 7, 8, 10, 13, 14, 15, 16])
def test_ntt_matches_oracle(k, log_n):
    n = 1 << log_n
    a = rand_scalars(n, 1000 + log_n)
    ctx = k.default_context(); lib = k._lib.load()
    for inverse in (0, 1):
        got = a.copy()
        assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(got), n, inverse) == 0
        assert np.array_equal(got, orc.fr_ntt(a, inverse=bool(inverse))), (log_n, inverse)


def test_ntt_roundtrip_2_20_and_linearity(k):
    n = 1 << 20
    rng = np.random.default_rng(3)
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)                # < 2^252 < r: canonical residues
    ctx = k.default_context(); lib = k._lib.load()
    f = a.copy()
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(f), n, 0) == 0
    back = f.copy()
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(back), n, 1) == 0
    assert np.array_equal(back, a)                     # eval <-> coeff round trip exact (BASELINE config 3)
    # spot check against the definition: F[i] = sum_j a_j w^(ij) at a few i, on a sparse input
    sparse = np.zeros((n, 4), np.uint64)
    idxs = [0, 1, 77777, n - 1]
    vals = [3, 5, 7, 11]
    for j, v in zip(idxs, vals):
        sparse[j] = pyref.fr_to_mont(v)
    g = sparse.copy()
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(g), n, 0) == 0
    w = pyref.root_of_unity(20)
    for i in (0, 1, 2, 12345, n // 2, n - 1):
        want = sum(v * pow(w, i * j, R_) for j, v in zip(idxs, vals)) % R_
        assert pyref.fr_from_mont(g[i]) == want


def test_ntt_errors(k):
    ctx = k.default_context(); lib = k._lib.load()
    a = rand_scalars(3, 1)
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(a), 3, 0) == k._lib.ERR_NOT_POWER_OF_TWO
    with pytest.raises(k.errors.PolynomialFFTError):
        p = k.PolynomialCoeffForm.__new__(k.PolynomialCoeffForm)
        p._coeffs = a; p._len_underlying_blob_bytes = 96
        p.to_eval_form()


def test_polynomial_forms_roundtrip(k, gettysburg):
    """primitives/tests/polynomial_test.rs:47-65, :99-113."""
    blob = k.Blob.from_raw_data(gettysburg)
    ev = blob.to_polynomial_eval_form()
    assert len(ev) == 64 and ev.len_underlying_blob_bytes() == 48 * 32
    co = ev.to_coeff_form()
    assert np.array_equal(co.to_eval_form().evaluations(), ev.evaluations())
    assert np.array_equal(co.coeffs(), orc.fr_ntt(ev.evaluations(), inverse=True))
    assert co.len_underlying_blob_bytes() == ev.len_underlying_blob_bytes()


# ---------------------------------------------------------------------------------------------------------
# KZG surface: roots, barycentric evaluation, commitments, proofs
# ---------------------------------------------------------------------------------------------------------
def test_roots_of_unity(k):
    for nbytes in (1, 32, 33, 48 * 32, 4096 * 32, (1 << 16) * 32):
        roots = k.helpers.calculate_roots_of_unity(nbytes)
        rc, want = orc.calculate_roots_of_unity(nbytes)
        assert rc == len(roots) and np.array_equal(roots, want)
    with pytest.raises(k.errors.GenericError, match="Length of data after padding is 0"):
        k.helpers.calculate_roots_of_unity(0)
    with pytest.raises(k.errors.GenericError, match="not valid with respect to the SRS"):
        k.helpers.calculate_roots_of_unity((268435456 + 1) * 32)


def test_barycentric_eval(k, gettysburg):
    """prover/tests/kzg_test.rs:31-55: every domain point returns the stored evaluation; off-domain vs oracle."""
    blob = k.Blob.from_raw_data(gettysburg)
    poly = blob.to_polynomial_eval_form()
    kzg = k.KZG.new(); kzg.calculate_and_store_roots_of_unity(len(blob))
    for i in range(64):
        y = k.helpers.evaluate_polynomial_in_evaluation_form(poly, kzg.get_nth_root_of_unity(i))
        assert np.array_equal(y, poly.get_evalualtion(i))
    for z in (5, 123456789, R_ - 2):
        zz = pyref.fr_to_mont(z)
        rc, want = orc.evaluate_polynomial_in_evaluation_form(poly.evaluations(), zz)
        assert rc == 0 and np.array_equal(k.helpers.evaluate_polynomial_in_evaluation_form(poly, zz), want)


def test_commitments_gettysburg(k, ref_srs, test_srs_wire, gettysburg):
    """coeff-form commit == eval-form commit (kzg_test.rs:57-89) == oracle == SURVEY.md §4.3 value."""
    blob = k.Blob.from_raw_data(gettysburg)
    kzg = k.KZG.new()
    ev = blob.to_polynomial_eval_form()
    c_eval = kzg.commit_eval_form(ev, ref_srs)
    c_blob = kzg.commit_blob(blob, ref_srs)
    c_coeff = kzg.commit_coeff_form(ev.to_coeff_form(), ref_srs)
    assert np.array_equal(c_eval, c_blob) and np.array_equal(c_eval, c_coeff)
    rc, want = orc.commit_eval_form(test_srs_wire, ev.evaluations(), literal=True)
    assert rc == 0 and np.array_equal(c_eval, want)
    assert pyref.point_from_wire(c_eval) == (
        2961155957874067312593973807786254905069537311739090798303675273531563528369,
        159565752702690920280451512738307422982252330088949702406468210607852362941)


def test_commit_errors(k, ref_srs):
    kzg = k.KZG.new()
    big = k.PolynomialEvalForm(pyref.frs_to_mont([1] * 3001))
    with pytest.raises(k.errors.SrsCapacityExceeded):
        kzg.commit_eval_form(big, ref_srs)
    with pytest.raises(k.errors.SerializationError, match="polynomial length is not correct"):
        kzg.commit_coeff_form(k.PolynomialCoeffForm(pyref.frs_to_mont([1] * 3001)), ref_srs)
    zero = k.PolynomialEvalForm(pyref.frs_to_mont([0] * 64))
    assert not kzg.commit_eval_form(zero, ref_srs).any()               # zero blob -> identity commitment


def test_proofs_match_reference_golden_vectors(k, ref_srs, gettysburg):
    """All 40 rows of kzg.proof.eq.input (on-domain branch, kzg.rs:237-260)."""
    blob = k.Blob.from_raw_data(gettysburg)
    poly = blob.to_polynomial_eval_form()
    kzg = k.KZG.new(); kzg.calculate_and_store_roots_of_unity(len(blob))
    for line in open(os.path.join(GOLDEN, "kzg.proof.eq.input")):
        line = line.strip()
        if not line:
            continue
        idx, x, y = line.split(",")
        proof = kzg.compute_proof_with_known_z_fr_index(poly, int(idx), ref_srs)
        assert pyref.point_from_wire(proof) == (int(x), int(y)), idx


def test_proof_off_domain_and_guards(k, ref_srs, test_srs_wire, gettysburg):
    blob = k.Blob.from_raw_data(gettysburg)
    poly = blob.to_polynomial_eval_form()
    kzg = k.KZG.new(); kzg.calculate_and_store_roots_of_unity(len(blob))
    rc, roots = orc.calculate_roots_of_unity(len(blob))
    for z in (987654321987654321, 3, R_ - 1):
        zz = pyref.fr_to_mont(z)
        proof, y = kzg._compute_proof_impl(poly, zz, ref_srs, want_y=True)
        rc, want, want_y = orc.compute_proof(test_srs_wire, poly.evaluations(), roots, zz, literal=False)
        assert rc == 0 and np.array_equal(proof, want) and np.array_equal(y, want_y)
    # Fiat-Shamir blob proof (kzg.rs:288-309) against the oracle's transcript + proof
    commitment = kzg.commit_blob(blob, ref_srs)
    z = orc.compute_challenge(blob.data(), commitment)
    assert np.array_equal(k.helpers.compute_challenge(blob, commitment), z)
    rc, want, _ = orc.compute_proof(test_srs_wire, poly.evaluations(), roots, z, literal=False)
    assert np.array_equal(kzg.compute_blob_proof(blob, commitment, ref_srs), want)
    # guards
    kzg2 = k.KZG.new(); kzg2.calculate_and_store_roots_of_unity(32 * 32)
    with pytest.raises(k.errors.GenericError, match="inconsistent length between blob and root of unities"):
        kzg2.compute_proof(poly, pyref.fr_to_mont(5), ref_srs)
    with pytest.raises(k.errors.GenericError, match="Root of unity not found"):
        kzg.compute_proof_with_known_z_fr_index(poly, 64, ref_srs)


def test_random_blob_commit_and_proof_2_12(k, tau_srs):
    """Random 4096-element polynomial on the known-tau SRS: commitment == p(tau) G, proof == q(tau) G."""
    n = 1 << 12
    rnd = random.Random(4)
    evals = [rnd.randrange(R_) for _ in range(n)]
    poly = k.PolynomialEvalForm(pyref.frs_to_mont(evals))
    kzg = k.KZG.new(); kzg.calculate_and_store_roots_of_unity(n * 32)
    c = kzg.commit_eval_form(poly, tau_srs)
    coeffs = pyref.frs_from_mont(orc.fr_ntt(poly.evaluations(), inverse=True))
    ptau = pyref.poly_eval(coeffs, TAU)
    assert pyref.point_from_wire(c) == pyref.ec_mul(ptau, (1, 2))
    z = rnd.randrange(R_)
    proof, y = kzg._compute_proof_impl(poly, pyref.fr_to_mont(z), tau_srs, want_y=True)
    yv = pyref.poly_eval(coeffs, z)
    assert pyref.fr_from_mont(y) == yv
    qtau = (ptau - yv) * pow(TAU - z, -1, R_) % R_
    assert pyref.point_from_wire(proof) == pyref.ec_mul(qtau, (1, 2))


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 8, 11, 12, 13, 14, 15])
def test_proofs_every_inversion_chain_shape(k, tau_srs, log_n):
    """compute_proof_impl (kzg.rs:128-178, :237-260) on domains of 1 .. 2^15 points: the batch inversion of the denominators is a
    chain of levels whose shape depends on n (small kernel only / + the fused x4 last level / + x4 middle levels), off the domain
    and ON it (first, second, middle, last and three random indices: the zero denominator sits in a different lane and level each time;
    up to 4 096 points the host finds the index and the inverses come from the domain's table 1 / (w^k - 1)).  Expected values by
    big integers on the known-tau SRS: y = f^(z), proof = ((f^(tau) - y) / (tau - z)) G1 with f^ from the barycentric formula."""
    n = 1 << log_n
    rnd = random.Random(0xC0DE + log_n)
    evals = [rnd.randrange(R_) for _ in range(n)]
    w = pyref.root_of_unity(log_n) if log_n else 1
    roots, cur = [], 1
    for _ in range(n):
        roots.append(cur)
        cur = cur * w % R_

    def bary(x):
        if n == 1:
            return evals[0]
        dens = [(x - r) % R_ for r in roots]
        pre, acc = [], 1
        for d in dens:
            pre.append(acc)
            acc = acc * d % R_
        inv = pow(acc, -1, R_)
        tot = 0
        for i in range(n - 1, -1, -1):
            tot += evals[i] * roots[i] % R_ * (inv * pre[i] % R_)
            inv = inv * dens[i] % R_
        return tot % R_ * (pow(x, n, R_) - 1) % R_ * pow(n, -1, R_) % R_

    ftau = bary(TAU)
    poly = k.PolynomialEvalForm(pyref.frs_to_mont(evals))
    kzg = k.KZG.new(); kzg.calculate_and_store_roots_of_unity(n * 32)
    z_off = rnd.randrange(R_)
    cases = [(z_off, bary(z_off))] + [(roots[m], evals[m]) for m in sorted({0, n // 2, n - 1, 1 % n, rnd.randrange(n), rnd.randrange(n), rnd.randrange(n)})]
    for z, y_want in cases:
        proof, y = kzg._compute_proof_impl(poly, pyref.fr_to_mont(z), tau_srs, want_y=True)
        assert pyref.fr_from_mont(y) == y_want, (log_n, z == z_off)
        if z == TAU % R_:
            continue
        if n == 1:
            want_pt = None                                          # the quotient of a constant is zero
        else:
            want_pt = pyref.ec_mul((ftau - y_want) * pow(TAU - z, -1, R_) % R_, (1, 2))
            if (ftau - y_want) % R_ == 0:
                want_pt = None
        assert pyref.point_from_wire(proof) == want_pt, (log_n, z == z_off)


# ---------------------------------------------------------------------------------------------------------
# g1_ifft (kzg.rs:263-285)
# ---------------------------------------------------------------------------------------------------------
def test_g1_ifft_matches_lagrange_fixture_and_oracle(k, ref_srs, test_srs_wire, gettysburg):
    kzg = k.KZG.new()
    want = []
    for line in open(os.path.join(GOLDEN, "lagrangeG1SRS.txt")):
        line = line.strip()
        if line:
            x, y = line.split(",")[-2:]
            want.append((int(x), int(y)))
    got = kzg.g1_ifft(64, ref_srs)
    assert [pyref.point_from_wire(g) for g in got] == want            # all 64 points of the reference's fixture
    for n in (1, 2, 8, 256):
        rc, o = orc.g1_ifft(test_srs_wire, n)
        assert rc == 0 and np.array_equal(kzg.g1_ifft(n, ref_srs), o), n
    with pytest.raises(k.errors.FFTError, match="length provided is not a power of 2"):   # kzg_test.rs:131-161
        kzg.g1_ifft(15, ref_srs)
    with pytest.raises(k.errors.SrsCapacityExceeded):
        kzg.g1_ifft(4096, ref_srs)
    # the reference's literal eval-form commitment: MSM over the Lagrange bases == this library's IFFT + MSM
    blob = k.Blob.from_raw_data(gettysburg)
    ev = blob.to_polynomial_eval_form()
    assert np.array_equal(k.helpers.g1_lincomb(got, ev.evaluations()), kzg.commit_eval_form(ev, ref_srs))


# ---------------------------------------------------------------------------------------------------------
# blob codec + commit_blob on the device (helpers.rs:40-57, kzg.rs:182-185)
# ---------------------------------------------------------------------------------------------------------
def test_blob_to_fr_and_commit_blob(k, tau_srs, ref_srs, gettysburg):
    lib = k._lib.load(); ctx = k.default_context()
    data = open(os.path.join(GOLDEN, "blobs.txt"), "rb").read()               # 4096 gnark-generated elements
    for cut in (len(data), len(data) - 7, 33, 32, 1):                         # ragged tails are right-padded with zeros
        d = data[:cut]
        buf = np.frombuffer(d, dtype=np.uint8)
        n_out = C.c_size_t(0)
        lib.kzg_blob_to_fr(ctx.handle, None, len(d), None, 0, C.byref(n_out))
        out = np.zeros((n_out.value, 4), np.uint64)
        assert lib.kzg_blob_to_fr(ctx.handle, buf.ctypes.data_as(k._lib.u8p), len(d), k._lib.ptr(out), len(out), C.byref(n_out)) == 0
        want = orc.to_fr_array(d)
        assert n_out.value == pyref.next_pow2(len(want)) and np.array_equal(out[:len(want)], want) and not out[len(want):].any()
    # non-canonical chunks are reduced mod r (From<Vec<u8>> for Blob does not validate)
    raw = bytes([0xFF] * 64)
    buf = np.frombuffer(raw, dtype=np.uint8); out = np.zeros((2, 4), np.uint64); n_out = C.c_size_t(0)
    assert lib.kzg_blob_to_fr(ctx.handle, buf.ctypes.data_as(k._lib.u8p), 64, k._lib.ptr(out), 2, C.byref(n_out)) == 0
    assert pyref.frs_from_mont(out) == [(2 ** 256 - 1) % R_] * 2
    # python mirror picks the device path for >= 4096 elements
    assert np.array_equal(k.helpers.to_fr_array(data), orc.to_fr_array(data))
    # commit_blob: bytes in, point out
    kzg = k.KZG.new()
    blob = k.Blob.from_raw_data(gettysburg)
    assert np.array_equal(kzg.commit_blob(blob, ref_srs), kzg.commit_eval_form(blob.to_polynomial_eval_form(), ref_srs))
    big = k.Blob.from_padded_unchecked(data)                                  # 4096 elements on the 2^16 known-tau SRS
    c = kzg.commit_blob(big, tau_srs)
    rc, want = orc.commit_eval_form(tau_srs.g1[:4096], orc.to_fr_array(data), literal=False)
    assert rc == 0 and np.array_equal(c, want)
    with pytest.raises(k.errors.SrsCapacityExceeded):
        kzg.commit_blob(k.Blob.from_padded_unchecked(bytes(32 * 3001)), ref_srs)
    zero = k.Blob.from_padded_unchecked(bytes(32 * 64))
    assert not kzg.commit_blob(zero, ref_srs).any()                           # zero blob -> identity (tests.rs:239-269)


def test_batched_lincomb_like_batch_verification(k, test_srs_wire):
    """The three equal-length MSMs of verifier/src/batch.rs:228,245,246 in one launch sequence (BASELINE config 5 shape),
    with duplicated and identity points (verifier/tests/tests.rs:343-346, :271-311)."""
    n = 1000
    proofs = test_srs_wire[:n].copy(); proofs[5] = proofs[6]; proofs[17] = 0
    c_minus_y = test_srs_wire[1000:1000 + n].copy(); c_minus_y[3] = c_minus_y[4]
    r_powers = rand_scalars(n, 31); r_times_z = rand_scalars(n, 32)
    got = k.helpers.g1_lincomb_batch([proofs, proofs, c_minus_y], [r_powers, r_times_z, r_powers])
    want = [orc.msm_pippenger(proofs, r_powers), orc.msm_pippenger(proofs, r_times_z), orc.msm_pippenger(c_minus_y, r_powers)]
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert np.array_equal(got[0], k.helpers.g1_lincomb(proofs, r_powers))
    one = k.helpers.g1_lincomb_batch([proofs], [r_powers])
    assert np.array_equal(one[0], want[0])
    with pytest.raises(k.errors.MsmError):
        k.helpers.g1_lincomb_batch([proofs, proofs[:10]], [r_powers, r_powers])


# ---------------------------------------------------------------------------------------------------------
# SRS::new: GPU decompression of gnark-format files (srs.rs:35-49, helpers.rs:175-226)
# ---------------------------------------------------------------------------------------------------------
def test_srs_new_decompresses_reference_file(k, test_srs_wire, gettysburg, tmp_path):
    path = os.path.join(GOLDEN, "g1.point")
    srs = k.SRS.new(path, 3000, 3000)
    assert len(srs) == 3000 and srs.order == 3000
    assert np.array_equal(srs.g1, test_srs_wire)                      # all 3000 points == srs.g1.points.string
    part = k.SRS.new(path, 3000, 64)
    assert np.array_equal(part.g1, test_srs_wire[:64])
    # a commitment over the file-loaded SRS equals the one over the uploaded SRS
    blob = k.Blob.from_raw_data(gettysburg)
    kzg = k.KZG.new()
    assert pyref.point_from_wire(kzg.commit_blob(blob, srs)) == (
        2961155957874067312593973807786254905069537311739090798303675273531563528369,
        159565752702690920280451512738307422982252330088949702406468210607852362941)
    with pytest.raises(k.errors.GenericError, match="Number of points to load exceeds SRS order."):   # kzg_test.rs:19-28
        k.SRS.new(path, 3000, 3001)
    raw = bytearray(open(path, "rb").read()[:32 * 8])
    # infinity encodings and flipped sign flags
    good = bytes(raw)
    raw[32:64] = bytes([0x40]) + bytes(31)                             # point 1 := infinity
    raw[64] ^= 0x40                                                    # point 2: smaller <-> larger y
    f = tmp_path / "mod.point"; f.write_bytes(bytes(raw))
    mod = k.SRS.new(str(f), 8, 8).g1
    assert not mod[1].any()
    assert np.array_equal(mod[2], orc.g1_neg(test_srs_wire[2])) and np.array_equal(mod[3], test_srs_wire[3])
    for i in range(8):
        rc, want = orc.g1_decompress_be(bytes(raw[32 * i:32 * i + 32]))
        assert rc == 0 and np.array_equal(mod[i], want)
    bad = bytearray(good); bad[32] = 0x41; bad[33:64] = bytes(31)      # infinity flag with stray bits
    f.write_bytes(bytes(bad))
    with pytest.raises(k.errors.DeserializationError, match="point at infinity not coded properly for g1"):
        k.SRS.new(str(f), 8, 8)
    x = 1
    while pow((x ** 3 + 3) % P, (P - 1) // 2, P) == 1:
        x += 1
    bad = bytearray(good); bad[96:128] = bytes([0x80]) + x.to_bytes(31, "big")          # x with x^3 + 3 a non-residue
    f.write_bytes(bytes(bad))
    with pytest.raises(k.errors.NotOnCurveError, match="compressed g1 point not on curve"):
        k.SRS.new(str(f), 8, 8)


# ---------------------------------------------------------------------------------------------------------
# randomized differential test: sizes, scalar patterns, window / segment settings
# ---------------------------------------------------------------------------------------------------------
def _patterned_scalars(rnd, n):
    kind = rnd.randrange(7)
    if kind == 0:
        vals = [rnd.randrange(R_) for _ in range(n)]                         # uniform
    elif kind == 1:
        vals = [rnd.randrange(1 << rnd.choice([1, 8, 31, 64, 128, 200])) for _ in range(n)]   # short scalars
    elif kind == 2:
        pool = [rnd.randrange(R_) for _ in range(3)]
        vals = [rnd.choice(pool) for _ in range(n)]                          # few distinct values -> heavy buckets
    elif kind == 3:
        vals = [R_ - 1 - rnd.randrange(1 << 16) for _ in range(n)]           # near r: carries through every window
    elif kind == 4:
        vals = [int.from_bytes(bytes(rnd.randrange(32, 127) for _ in range(31)), "big") for _ in range(n)]   # blob-like
    elif kind == 5:
        vals = [(1 << rnd.randrange(254)) % R_ for _ in range(n)]           # one-hot bits
    else:
        vals = [0 if rnd.random() < 0.7 else rnd.randrange(R_) for _ in range(n)]   # mostly zero (padded polynomials)
    return pyref.frs_to_mont(vals)


def test_msm_randomized_differential(k, ref_srs, tau_srs, test_srs_wire):
    rnd = random.Random(20261003)
    tau_pts = tau_srs.g1[:6000]
    for case in range(36):
        use_tau = case % 3 == 0
        srs, pts = (tau_srs, tau_pts) if use_tau else (ref_srs, test_srs_wire)
        nmax = 6000 if use_tau else 3000
        n = rnd.choice([1, 2, 63, 64, 65, 127, 128, 129, rnd.randrange(1, nmax), rnd.randrange(1, nmax), nmax])
        off = rnd.randrange(0, nmax - n + 1)
        sc = _patterned_scalars(rnd, n)
        c = rnd.choice([0, 0, 0, 4, 7, 9, 12, 15, 16])                       # 0 = table mode (SRS tables), else generic
        seg = rnd.choice([0, 0, 1, 3, 8, 33, 200])
        srs.ctx.set_msm_window(c, seg)
        try:
            got = msm_srs(k, srs, sc, offset=off)
        finally:
            srs.ctx.set_msm_window(0, 0)
        want = orc.msm_pippenger(pts[off:off + n], sc)
        assert np.array_equal(got, want), (case, n, off, c, seg)


def test_msm_begin_end_pipeline(k, tau_srs, ref_srs, test_srs_wire):
    """kzg_msm_g1_srs_device_begin / kzg_msm_g1_srs_end: two MSMs in flight on the two slots give exactly the results of
    the synchronous call; slot misuse is rejected; ShardedMsm.commit_stream yields the commitments in order."""
    import torch
    from rust_kzg_bn254_amd.sharding import ShardedMsm
    lib = k._lib.load()
    ctx = tau_srs.ctx
    n = 1 << 14
    bufs = [rand_scalars(n, 900 + i) for i in range(5)]
    want = [msm_srs(k, tau_srs, b) for b in bufs]
    dev = [torch.from_numpy(np.ascontiguousarray(b).view(np.int64)).cuda() for b in bufs]
    torch.cuda.synchronize()
    sh = ShardedMsm(ctx, n)
    for depth in (None, 1, 2, 3, 4):
        got = list(sh.commit_stream(tau_srs, [d.data_ptr() for d in dev], depth=depth))
        assert len(got) == 5 and all(np.array_equal(g, w) for g, w in zip(got, want)), depth
    # both slots busy at once, ended in the opposite order; partial (XYZZ) output folds to the same point
    assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, tau_srs.handle, 0, C.c_void_p(dev[0].data_ptr()), n, 0) == 0
    assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, tau_srs.handle, 0, C.c_void_p(dev[1].data_ptr()), n, 1) == 0
    assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, tau_srs.handle, 0, C.c_void_p(dev[2].data_ptr()), n, 1) == k._lib.ERR_INVALID_ARG
    # a synchronous call would reuse slot 0's workspace: refused while slot 0 is in flight
    o8 = np.zeros(8, np.uint64); i8 = C.c_uint8(0)
    assert lib.kzg_msm_g1_srs_device(ctx.handle, tau_srs.handle, 0, C.c_void_p(dev[2].data_ptr()), n, k._lib.ptr(o8), C.byref(i8)) == k._lib.ERR_INVALID_ARG
    assert "slot 0" in ctx.last_error()
    part = np.zeros(16, np.uint64)
    assert lib.kzg_msm_g1_srs_end(ctx.handle, 1, None, None, k._lib.ptr(part)) == 0
    assert np.array_equal(k.sharding.fold_partials(part.reshape(1, 16)), want[1])
    out = np.zeros(8, np.uint64); inf = C.c_uint8(9)
    assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, k._lib.ptr(out), C.byref(inf), None) == 0
    assert np.array_equal(out, want[0]) and inf.value == 0
    assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, k._lib.ptr(out), C.byref(inf), None) == k._lib.ERR_INVALID_ARG   # idle slot
    assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, tau_srs.handle, 0, C.c_void_p(dev[0].data_ptr()), n, k._lib.NUM_SLOTS) == k._lib.ERR_INVALID_ARG
    assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, tau_srs.handle, (1 << 16) - 5, C.c_void_p(dev[0].data_ptr()), n, 0) == k._lib.ERR_MSM_LENGTH_MISMATCH
    # small generic-mode SRS (no tables) through the same path, against the oracle
    m = 700
    sc = rand_scalars(m, 4242)
    d = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).cuda(); torch.cuda.synchronize()
    assert lib.kzg_msm_g1_srs_device_begin(ref_srs.ctx.handle, ref_srs.handle, 0, C.c_void_p(d.data_ptr()), m, 1) == 0
    assert lib.kzg_msm_g1_srs_end(ref_srs.ctx.handle, 1, k._lib.ptr(out), C.byref(inf), None) == 0
    assert np.array_equal(out, orc.msm_pippenger(test_srs_wire[:m], sc))


def test_commit_coeff_form_stream_host_buffers(k, tau_srs):
    """KZG.commit_coeff_form_stream (kzg_msm_g1_srs_begin/_end with host scalars): same points as commit_coeff_form, in order,
    for polynomials of different lengths."""
    kz = k.KZG.new()
    rnd = random.Random(31)
    polys = [k.PolynomialCoeffForm(pyref.frs_to_mont([rnd.randrange(R_) for _ in range(n)])) for n in (4096, 1, 333, 65536, 2048, 7)]
    want = [kz.commit_coeff_form(p, tau_srs) for p in polys]
    got = list(kz.commit_coeff_form_stream(polys, tau_srs))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert list(kz.commit_coeff_form_stream([], tau_srs)) == []


def test_sharded_commit_and_proof_partials(k, tau_srs):
    """kzg_commit_eval_form_partial / kzg_compute_proof_partial (config 4): three SRS shards [0,a), [a,b), [b,n) generated with
    first_power, each commits its slice; the folded partials are checked against BIG-INTEGER values on the known-tau SRS
    (commit == f^(tau) G, proof == ((f^(tau) - y) / (tau - z)) G), not against another run of the HIP path."""
    from rust_kzg_bn254_amd.sharding import fold_partials
    lib = k._lib.load()
    ctx = tau_srs.ctx
    n = 1 << 13
    rnd = random.Random(2024)
    vals = [rnd.randrange(R_) for _ in range(n)]
    evals = pyref.frs_to_mont(vals)
    poly = k.PolynomialEvalForm(evals)
    # big-integer ground truth: barycentric evaluation of the evaluation-form polynomial at tau and at z
    w = pyref.root_of_unity(13)
    roots, cur = [], 1
    for _ in range(n):
        roots.append(cur); cur = cur * w % R_
    def bary(x):
        s_ = sum(f * r % R_ * pow(x - r, -1, R_) for f, r in zip(vals, roots)) % R_
        return s_ * (pow(x, n, R_) - 1) % R_ * pow(n, -1, R_) % R_
    zi = 0x1234567890ABCDEF1234567
    ftau, yz = bary(TAU), bary(zi)
    want_c = pyref.ec_mul(ftau, (1, 2))
    want_p = pyref.ec_mul((ftau - yz) * pow(TAU - zi, -1, R_) % R_, (1, 2))
    kz = k.KZG.new()
    kz.calculate_and_store_roots_of_unity(n * 32)
    z = k.fr.fr_from_int(zi)
    bounds = [0, 3000, 3001, n]                       # uneven shards, one of a single point
    parts_c, parts_p, ys = [], [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        shard = k.SRS.generate(TAU, hi - lo, first_power=lo, ctx=ctx)
        pc = np.zeros(16, np.uint64); pp = np.zeros(16, np.uint64); y = np.zeros(4, np.uint64)
        ev = np.ascontiguousarray(evals)
        assert lib.kzg_commit_eval_form_partial(ctx.handle, shard.handle, lo, k._lib.ptr(ev), n, k._lib.ptr(pc)) == 0
        assert lib.kzg_compute_proof_partial(ctx.handle, shard.handle, lo, k._lib.ptr(ev), n, None, n, k._lib.ptr(np.ascontiguousarray(z)),
                                             k._lib.ptr(pp), k._lib.ptr(y)) == 0
        parts_c.append(pc); parts_p.append(pp); ys.append(y)
        shard.close()
    assert pyref.point_from_wire(fold_partials(np.stack(parts_c))) == want_c
    assert pyref.point_from_wire(fold_partials(np.stack(parts_p))) == want_p
    assert all(pyref.fr_from_mont(y) == yz for y in ys)
    # the single-GPU calls give the same points
    assert pyref.point_from_wire(kz.commit_eval_form(poly, tau_srs)) == want_c
    assert pyref.point_from_wire(kz.compute_proof(poly, z, tau_srs)) == want_p
    # a shard that starts beyond the polynomial contributes the identity
    shard = k.SRS.generate(TAU, 16, first_power=n, ctx=ctx)
    pc = np.ones(16, np.uint64)
    assert lib.kzg_commit_eval_form_partial(ctx.handle, shard.handle, n, k._lib.ptr(np.ascontiguousarray(evals)), n, k._lib.ptr(pc)) == 0
    assert not pc.any()
    # ShardedKzg with world = 1 is the plain call
    from rust_kzg_bn254_amd.sharding import ShardedKzg
    sk = ShardedKzg(ctx, tau_srs, n)
    assert pyref.point_from_wire(sk.commit_eval_form(poly)) == want_c
    assert pyref.point_from_wire(sk.compute_proof(poly, z)) == want_p


def test_msm_2_20_edge_sets(k):
    """SURVEY.md §8d edge sets at the full BASELINE size (2^20), checked by size-independent properties on a known-tau SRS:
    uniform scalars; all = r-1; one-hot; 2^20 copies of one point with scalar 1 (P+P everywhere, one bucket holds everything);
    (P, -P) pairs with equal scalars (identity inside every bucket)."""
    n = 1 << 20
    srs = k.SRS.generate(TAU, n)
    G = (1, 2)
    rng = np.random.default_rng(2020)
    # uniform in [0, r): rejection-free construction from 4 limbs reduced mod r
    raw = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    vals = [((int(a) << 189) ^ (int(b) << 126) ^ (int(c) << 63) ^ int(d)) % R_ for a, b, c, d in raw]
    got = msm_srs(k, srs, pyref.frs_to_mont(vals))
    ptau, cur = 0, 1
    for v in vals:
        ptau = (ptau + v * cur) % R_
        cur = cur * TAU % R_
    assert pyref.point_from_wire(got) == pyref.ec_mul(ptau, G)
    geo = (pow(TAU, n, R_) - 1) * pow(TAU - 1, -1, R_) % R_
    rm1 = np.ascontiguousarray(np.broadcast_to(pyref.fr_to_mont(R_ - 1), (n, 4)))
    assert pyref.point_from_wire(msm_srs(k, srs, rm1)) == pyref.ec_mul((-geo) % R_, G)
    onehot = np.zeros((n, 4), np.uint64)
    onehot[777777] = pyref.fr_to_mont(0xABCDEF0123456789ABCDEF)
    assert pyref.point_from_wire(msm_srs(k, srs, onehot)) == pyref.ec_mul(0xABCDEF0123456789ABCDEF * pow(TAU, 777777, R_) % R_, G)
    assert not msm_srs(k, srs, np.zeros((n, 4), np.uint64)).any()
    srs.close()
    # caller bases (generic mode): one point 2^20 times, scalars 1  ->  2^20 * P
    p7 = pyref.ec_mul(7, G)
    dup = np.ascontiguousarray(np.broadcast_to(pyref.point_to_wire(p7), (n, 8)))
    ones = np.ascontiguousarray(np.broadcast_to(pyref.fr_to_mont(1), (n, 4)))
    assert pyref.point_from_wire(k.helpers.g1_lincomb(dup, ones)) == pyref.ec_mul(7 * n % R_, G)
    # (P, -P) pairs with equal scalars -> identity
    pm = dup.copy()
    pm[1::2] = pyref.point_to_wire((p7[0], (-p7[1]) % P))
    sc = np.repeat(pyref.frs_to_mont(vals[: n // 2]), 2, axis=0)
    assert not k.helpers.g1_lincomb(pm, sc).any()


def test_commit_eval_and_blob_streams(k, tau_srs, gettysburg):
    """kzg_commit_eval_form_begin / kzg_commit_blob_begin + kzg_msm_g1_srs_end: streamed commitments equal the synchronous ones."""
    kz = k.KZG.new()
    rnd = random.Random(77)
    blobs = [k.Blob.from_raw_data(bytes(rnd.randrange(32, 127) for _ in range(m))) for m in (31 * 4096, 40, 31 * 1000 + 5, 31 * 65536, 1)]
    blobs.append(k.Blob.from_raw_data(gettysburg))
    want = [kz.commit_blob(b, tau_srs) for b in blobs]
    assert all(np.array_equal(g, w) for g, w in zip(kz.commit_blob_stream(blobs, tau_srs), want))
    polys = [b.to_polynomial_eval_form() for b in blobs]
    want_e = [kz.commit_eval_form(p, tau_srs) for p in polys]
    assert all(np.array_equal(a, b) for a, b in zip(want, want_e))
    got_e = list(kz.commit_eval_form_stream(polys, tau_srs))
    assert len(got_e) == len(want_e) and all(np.array_equal(g, w) for g, w in zip(got_e, want_e))
    # slot already in flight -> rejected; SRS too short -> SrsCapacityExceeded
    lib = k._lib.load(); ctx = tau_srs.ctx
    ev = np.ascontiguousarray(polys[0].evaluations())
    assert lib.kzg_commit_eval_form_begin(ctx.handle, tau_srs.handle, k._lib.ptr(ev), len(ev), 0) == 0
    assert lib.kzg_commit_eval_form_begin(ctx.handle, tau_srs.handle, k._lib.ptr(ev), len(ev), 0) == k._lib.ERR_INVALID_ARG
    out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
    assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, k._lib.ptr(out), C.byref(inf), None) == 0
    assert np.array_equal(out, want[0])
    big = np.zeros((1 << 17, 4), np.uint64)
    assert lib.kzg_commit_eval_form_begin(ctx.handle, tau_srs.handle, k._lib.ptr(big), len(big), 1) == k._lib.ERR_SRS_CAPACITY_EXCEEDED
    assert lib.kzg_commit_eval_form_begin(ctx.handle, tau_srs.handle, k._lib.ptr(big), 3, 1) == k._lib.ERR_NOT_POWER_OF_TWO


def test_compute_proof_stream(k, tau_srs, ref_srs, gettysburg):
    """kzg_compute_proof_begin / _end: streamed proofs (off-domain and on-domain z, different sizes) equal the synchronous call,
    y included; the golden on-domain proofs of kzg.proof.eq.input come out of the stream too."""
    kz = k.KZG.new()
    rnd = random.Random(123)
    items = []
    for n in (64, 4096, 1 << 15, 512, 1 << 16):
        poly = k.PolynomialEvalForm(pyref.frs_to_mont([rnd.randrange(R_) for _ in range(n)]))
        items.append((poly, k.fr.fr_from_int(rnd.randrange(R_))))
    # an on-domain point: z = w^5 of the 4096 domain
    w = pyref.root_of_unity(12)
    items.append((items[1][0], k.fr.fr_from_int(pow(w, 5, R_))))
    want = []
    for p, z in items:
        kz.calculate_and_store_roots_of_unity(len(p) * 32)
        want.append(kz._compute_proof_impl(p, z, tau_srs, want_y=True))
    got = list(kz.compute_proof_stream(items, tau_srs, want_y=True))
    assert len(got) == len(want)
    for (gp, gy), (wp, wy) in zip(got, want):
        assert np.array_equal(gp, wp) and np.array_equal(gy, wy)
    assert np.array_equal(got[-1][1], items[1][0].evaluations()[5])          # on-domain: y = f_5
    # reference golden vectors through the stream (SRS = first 64 Lagrange points is not what they used: compute via ref_srs)
    blob = k.Blob.from_raw_data(gettysburg)
    poly = blob.to_polynomial_eval_form()
    kz.calculate_and_store_roots_of_unity(len(blob))
    rows = []
    for line in open(os.path.join(GOLDEN, "kzg.proof.eq.input")):
        if line.strip():
            idx, x, y = line.strip().split(",")
            rows.append((int(idx), (int(x), int(y))))
    sel = rows[:8]
    proofs = list(kz.compute_proof_stream([(poly, kz.get_nth_root_of_unity(i)) for i, _ in sel], ref_srs))
    for (i, pt), pr in zip(sel, proofs):
        assert pyref.point_from_wire(pr) == pt, i


def test_slots_mixed_stress(k, tau_srs):
    """All KZG_NUM_SLOTS slots busy with a random mix of asynchronous calls (MSM from device / host scalars, eval-form and blob
    commitments, proofs) of random sizes, ended in random order: every result equals the synchronous call's."""
    import torch
    lib = k._lib.load(); ctx = tau_srs.ctx
    rnd = random.Random(2026)
    kz = k.KZG.new()
    u8p = k._lib.u8p
    jobs = []
    for j in range(40):
        kind = rnd.choice(["msm_dev", "msm_host", "eval", "blob", "proof"])
        n = 1 << rnd.randrange(0, 15)
        if kind == "blob":
            raw = bytes(rnd.randrange(32, 127) for _ in range(rnd.randrange(1, 31 * 4096)))
            blob = k.Blob.from_raw_data(raw)
            want = kz.commit_blob(blob, tau_srs)
            jobs.append((kind, np.frombuffer(blob.data(), dtype=np.uint8).copy(), None, want, None))
            continue
        sc = rand_scalars(n, 5000 + j)
        if kind in ("msm_dev", "msm_host"):
            m = rnd.randrange(1, n + 1)
            sc = np.ascontiguousarray(sc[:m])
            want = msm_srs(k, tau_srs, sc)
            dev = torch.from_numpy(sc.view(np.int64)).cuda() if kind == "msm_dev" else None
            jobs.append((kind, sc, dev, want, None))
        elif kind == "eval":
            want = kz.commit_eval_form(k.PolynomialEvalForm(sc), tau_srs)
            jobs.append((kind, np.ascontiguousarray(sc), None, want, None))
        else:
            z = k.fr.fr_from_int(rnd.randrange(R_))
            kz.calculate_and_store_roots_of_unity(n * 32)
            wp, wy = kz._compute_proof_impl(k.PolynomialEvalForm(sc), z, tau_srs, want_y=True)
            jobs.append((kind, np.ascontiguousarray(sc), np.ascontiguousarray(z), wp, wy))
    torch.cuda.synchronize()
    busy = {}                                        # slot -> job index
    todo = list(range(len(jobs)))
    done = 0
    while todo or busy:
        free = [s for s in range(k._lib.NUM_SLOTS) if s not in busy]
        if todo and free and (not busy or rnd.random() < 0.7):
            s = rnd.choice(free); j = todo.pop(0)
            kind, data, aux, _, _ = jobs[j]
            if kind == "msm_dev":
                rc = lib.kzg_msm_g1_srs_device_begin(ctx.handle, tau_srs.handle, 0, C.c_void_p(aux.data_ptr()), len(data), s)
            elif kind == "msm_host":
                rc = lib.kzg_msm_g1_srs_begin(ctx.handle, tau_srs.handle, 0, k._lib.ptr(data), len(data), s)
            elif kind == "eval":
                rc = lib.kzg_commit_eval_form_begin(ctx.handle, tau_srs.handle, k._lib.ptr(data), len(data), s)
            elif kind == "blob":
                rc = lib.kzg_commit_blob_begin(ctx.handle, tau_srs.handle, data.ctypes.data_as(u8p), data.size, s)
            else:
                rc = lib.kzg_compute_proof_begin(ctx.handle, tau_srs.handle, k._lib.ptr(data), len(data), None, len(data), k._lib.ptr(aux), s)
            assert rc == 0, (kind, rc, ctx.last_error())
            busy[s] = j
        else:
            s = rnd.choice(list(busy)); j = busy.pop(s)
            kind, _, _, want, want_y = jobs[j]
            out = np.zeros(8, np.uint64); inf = C.c_uint8(0); y = np.zeros(4, np.uint64)
            if kind == "proof":
                assert lib.kzg_compute_proof_end(ctx.handle, s, k._lib.ptr(out), C.byref(inf), k._lib.ptr(y)) == 0
                assert np.array_equal(y, want_y), (j, kind)
            else:
                assert lib.kzg_msm_g1_srs_end(ctx.handle, s, k._lib.ptr(out), C.byref(inf), None) == 0
            assert np.array_equal(out, want), (j, kind)
            done += 1
    assert done == len(jobs)


def test_msm_adversarial_digit_patterns(k, tau_srs):
    """Signed-window recoding corner cases in table mode (one bucket set for all windows): every window digit at the signed
    boundary (2^(c-1): recoded to -2^(c-1) with a carry), all ones (carry chains through every window), the largest scalar
    r-1, alternating boundary / zero digits, single top-window digits -- for c = 12 (2^16-point SRS) and c = 17 (2^20-point SRS).
    Checked against sum_i s_i tau^i on the known-tau SRS (independent big-integer arithmetic)."""
    def patterns(c):
        full = (1 << 254) - 1
        half = sum(1 << (c * w + c - 1) for w in range(0, 254 // c + 1))
        alt = sum(1 << (c * w + c - 1) for w in range(0, 254 // c + 1, 2))
        return [half % R_, (half - 1) % R_, full % R_, R_ - 1, R_ - 2, alt % R_, 1 << 253, (1 << 253) - 1, (1 << (c * 3)) - 1, 1, 0]

    def check(srs, n, c):
        pats = patterns(c)
        vals = [pats[i % len(pats)] for i in range(n)]
        got = msm_srs(k, srs, pyref.frs_to_mont(vals))
        ptau, cur = 0, 1
        for v in vals:
            ptau = (ptau + v * cur) % R_
            cur = cur * TAU % R_
        assert pyref.point_from_wire(got) == pyref.ec_mul(ptau, (1, 2)), c
        # all scalars equal to the boundary pattern: every window has one bucket with n entries
        same = pyref.frs_to_mont([pats[0]] * n)
        geo = (pow(TAU, n, R_) - 1) * pow(TAU - 1, -1, R_) % R_
        assert pyref.point_from_wire(msm_srs(k, srs, same)) == pyref.ec_mul(pats[0] * geo % R_, (1, 2)), c

    check(tau_srs, 1 << 16, 12)
    big = k.SRS.generate(TAU, 1 << 20)
    check(big, 1 << 18, 17)
    big.close()


def test_tiny_msm_as_sums_of_per_bit_table_points(k, tau_srs):
    """MSMs of up to 4 096 pairs on an SRS with per-bit tables run as plain sums of table points (k_bitsum_level1 / 2: the non-adjacent
    form of every scalar, 8 / 16 / 32 positions per lane quad by length): every length around the workgroup / chunk boundaries, offsets into the SRS,
    and scalars that stress the recoding -- 0, 1, r - 1, 2^253, runs of ones (carries through every position), 0101.. / 1010.. / 0011..
    patterns, 3 k for such k -- against the closed form on the known-tau SRS."""
    rnd = random.Random(2048)
    def tau_check(vals, off=0):
        ptau, cur = 0, pow(TAU, off, R_)
        for v in vals:
            ptau = (ptau + v * cur) % R_
            cur = cur * TAU % R_
        return pyref.ec_mul(ptau, (1, 2)) if ptau else None
    ones = (1 << 253) - 1
    a5 = int("5" * 63, 16) % R_
    aa = int("a" * 62, 16) % R_
    c3 = int("3" * 63, 16) % R_
    special = [0, 1, 2, 3, R_ - 1, R_ - 2, 1 << 253, ones, ones - 1, (1 << 200) - 1, a5, aa, c3, 3 * a5 % R_, 3 * aa % R_, (R_ - 1) // 2, (R_ + 1) // 2,
               (1 << 8) - 1, 1 << 8, (1 << 248) + (1 << 247), sum(1 << (8 * j + 7) for j in range(31)), sum(3 << (8 * j + 6) for j in range(31))]
    for n in (1, 2, 3, 31, 32, 33, 63, 64, 65, 500, 512, 513, 1024, 1025, 2047, 2048, 2049, 4095, 4096):
        vals = [rnd.randrange(R_) for _ in range(n)]
        for j, sp in enumerate(special):
            if j < n:
                vals[(j * 7) % n] = sp
        got = msm_srs(k, tau_srs, pyref.frs_to_mont(vals))
        assert pyref.point_from_wire(got) == tau_check(vals), n
    for n, off in ((100, 1), (777, 5000), (2048, (1 << 16) - 2048), (4096, (1 << 16) - 4096), (1, (1 << 16) - 1)):
        vals = [rnd.randrange(R_) for _ in range(n)]
        assert pyref.point_from_wire(msm_srs(k, tau_srs, pyref.frs_to_mont(vals), offset=off)) == tau_check(vals, off), (n, off)
    for vals in ([0] * 300, [special[7]] * 2048, [R_ - 1] * 1000, [special[10]] * 4096):            # all zero (identity), all-ones scalars, r - 1 everywhere
        assert pyref.point_from_wire(msm_srs(k, tau_srs, pyref.frs_to_mont(vals))) == tau_check(vals)


def test_reduction_kernels_on_lane_pairs_and_lane_quads(k, ref_srs, small_srs, test_srs_wire, tau_srs):
    """The two forms of the table-mode reduction kernels (curve_pair.h: one point per two lanes, used beside another MSM in flight;
    curve_quad.h: per four lanes, used by an MSM that runs alone) forced one after the other over the same inputs: sparse MSMs (the
    fused first level, with its heavy-bucket path: equal scalars), mid-size ones (accumulate kernel + both levels), the NAF mode, and the
    boundary digit patterns.  Each against the oracle or the known-tau closed form."""
    ctx = ref_srs.ctx
    assert tau_srs.ctx is ctx
    assert small_srs.ctx is ctx
    def tau_check(vals):
        ptau, cur = 0, 1
        for v in vals:
            ptau = (ptau + v * cur) % R_
            cur = cur * TAU % R_
        return pyref.ec_mul(ptau, (1, 2))
    rnd = random.Random(77)
    try:
        for lanes in (2, 4):
            ctx.set_reduction_lanes(lanes)
            for n in (1, 2, 33, 700, 1500):                     # (an SRS below 2^11 points has no per-bit tables: these go through the buckets)
                sc = rand_scalars(n, 400 + n)
                assert np.array_equal(msm_srs(k, small_srs, sc), orc.msm_pippenger(test_srs_wire[:n], sc)), (lanes, n)
            same = np.ascontiguousarray(np.broadcast_to(rand_scalars(1, 5), (600, 4))).copy()      # 600 entries in one bucket per window
            assert np.array_equal(msm_srs(k, small_srs, same), orc.msm_pippenger(test_srs_wire[:600], same)), lanes
            sc = rand_scalars(3000, 3400)                       # (bit sums on the 3 000-point SRS: the same result under either setting)
            assert np.array_equal(msm_srs(k, ref_srs, sc), orc.msm_pippenger(test_srs_wire[:3000], sc)), lanes
            for n in (5000, (1 << 14) + 5, 1 << 16):
                vals = [rnd.randrange(R_) for _ in range(n)]
                vals[::7] = [R_ - 1] * len(vals[::7])
                vals[3::11] = [(1 << 253) - 1] * len(vals[3::11])
                assert pyref.point_from_wire(msm_srs(k, tau_srs, pyref.frs_to_mont(vals))) == tau_check(vals), (lanes, n)
            few = [rnd.randrange(R_) for _ in range(3)]
            vals = [few[i % 3] for i in range(40000)]                                              # three heavy buckets per window, NAF mode
            assert pyref.point_from_wire(msm_srs(k, tau_srs, pyref.frs_to_mont(vals))) == tau_check(vals), lanes
    finally:
        ctx.set_reduction_lanes(0)
    with pytest.raises(ValueError):
        ctx.set_reduction_lanes(3)


@pytest.mark.parametrize("log_n", [14, 20, 22])
def test_ntt_extreme_values(k, log_n):
    """Magnitude corner cases of the lazy-reduction NTT (values grow by 2m per butterfly stage before the next multiply):
    a constant vector of r-1 (every butterfly adds two maximal values; transform = n (r-1) at index 0, zero elsewhere), the
    alternating vector (r-1, 1, r-1, 1, ..) (energy at indices 0 and n/2), and a round trip of a vector of maximal canonical values."""
    n = 1 << log_n
    ctx = k.default_context(); lib = k._lib.load()
    top = pyref.fr_to_mont(R_ - 1)
    a = np.ascontiguousarray(np.broadcast_to(top, (n, 4))).copy()
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(a), n, 0) == 0
    assert pyref.fr_from_mont(a[0]) == n * (R_ - 1) % R_
    assert not a[1:].any()
    b = np.ascontiguousarray(np.broadcast_to(top, (n, 4))).copy()
    b[1::2] = pyref.fr_to_mont(1)
    orig = b.copy()
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(b), n, 0) == 0
    half = n // 2
    assert pyref.fr_from_mont(b[0]) == half * ((R_ - 1) + 1) % R_            # = 0
    assert pyref.fr_from_mont(b[half]) == half * ((R_ - 1) - 1) % R_
    mask = np.ones(n, bool); mask[[0, half]] = False
    assert not b[mask].any()
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(b), n, 1) == 0
    assert np.array_equal(b, orig)
    rng = np.random.default_rng(log_n)
    vals = [R_ - 1 - int(x) for x in rng.integers(0, 1000, size=64)]
    c = np.ascontiguousarray(np.tile(pyref.frs_to_mont(vals), (n // 64, 1)))
    orig = c.copy()
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(c), n, 1) == 0
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(c), n, 0) == 0
    assert np.array_equal(c, orig)


def test_stream_error_does_not_poison_the_context(k, tau_srs):
    """an item that fails mid-stream (here: a polynomial longer than the SRS) while another one is in flight, or a
    consumer that stops early, must leave no slot pending: the next synchronous call on the same context works."""
    kz = k.KZG.new()
    rnd = random.Random(99)
    ok = k.PolynomialEvalForm(pyref.frs_to_mont([rnd.randrange(R_) for _ in range(1024)]))
    too_long = k.PolynomialEvalForm(np.zeros((1 << 17, 4), np.uint64))
    want = kz.commit_eval_form(ok, tau_srs)
    with pytest.raises(k.errors.SrsCapacityExceeded):
        list(kz.commit_eval_form_stream([ok, too_long, ok], tau_srs))
    assert np.array_equal(kz.commit_eval_form(ok, tau_srs), want)             # slot 0 was drained
    gen = kz.commit_eval_form_stream([ok, ok, ok, ok], tau_srs)
    assert np.array_equal(next(gen), want)
    gen.close()                                                               # consumer stops with one commitment in flight
    assert np.array_equal(kz.commit_eval_form(ok, tau_srs), want)
    co = ok.to_coeff_form()
    with pytest.raises(k.errors.SerializationError):
        list(kz.commit_coeff_form_stream([co, k.PolynomialCoeffForm(np.zeros((1 << 17, 4), np.uint64))], tau_srs))
    assert np.array_equal(kz.commit_coeff_form(co, tau_srs), want)
    z = k.fr.fr_from_int(12345)
    kz.calculate_and_store_roots_of_unity(1024 * 32)
    wantp = kz.compute_proof(ok, z, tau_srs)
    with pytest.raises(k.errors.SrsCapacityExceeded):
        list(kz.compute_proof_stream([(ok, z), (too_long, z)], tau_srs))
    assert np.array_equal(kz.compute_proof(ok, z, tau_srs), wantp)
    # ShardedMsm.commit_stream: consumer stops early
    import torch
    from rust_kzg_bn254_amd.sharding import ShardedMsm
    d = torch.from_numpy(np.ascontiguousarray(co.coeffs()).view(np.int64)).cuda(); torch.cuda.synchronize()
    sh = ShardedMsm(tau_srs.ctx, 1024)
    g2 = sh.commit_stream(tau_srs, [d.data_ptr()] * 6, depth=3)
    assert np.array_equal(next(g2), want)
    g2.close()
    assert np.array_equal(kz.commit_coeff_form(co, tau_srs), want)


def test_msm_batch_of_64_small_msms(k, test_srs_wire):
    """kzg_msm_g1_batch advertises batch <= 64; 64 MSMs of 8 pairs (c = 4: 64 windows each) fit one launch."""
    n, batch = 8, 64
    rnd = random.Random(64)
    pts = [test_srs_wire[rnd.randrange(3000 - n):][:n].copy() for _ in range(batch)]
    scs = [rand_scalars(n, 6400 + i) for i in range(batch)]
    got = k.helpers.g1_lincomb_batch(pts, scs)
    for g, p, s_ in zip(got, pts, scs):
        assert np.array_equal(g, orc.msm_pippenger(p, s_))


def test_g1_ifft_paths_and_lagrange_cache(k, tau_srs, ref_srs, test_srs_wire):
    """g1_ifft beyond the fixture: the high-radix direct stages (n <= 2^14) and the radix-2 butterflies (n = 2^15) against the oracle
    at n = 4 .. 2048 and the closed form l_i(tau) G at 2^15; the Lagrange basis
    kept on the device (kzg_srs_cache_lagrange / kzg_srs_lagrange): commit_eval_form over it == IFFT + MSM == oracle."""
    kzg = k.KZG.new()
    rc, want = orc.g1_ifft(test_srs_wire, 1024)
    assert rc == 0
    got = kzg.g1_ifft(1024, ref_srs)
    assert np.array_equal(got, want)
    for n in (4, 32, 512, 2048):                                     # the staged kernels (this 3000-point SRS has per-bit tables: 64 .. 2048 go through them)
        rc, want_n = orc.g1_ifft(test_srs_wire, n)
        assert rc == 0 and np.array_equal(kzg.g1_ifft(n, ref_srs), want_n), n
    # known tau: L_i = l_i(tau) G with l_i the Lagrange polynomial of the domain, checked at a few i for n = 2^15
    n = 1 << 15
    L = kzg.g1_ifft(n, tau_srs)
    w = pyref.root_of_unity(15)
    zn = (pow(TAU, n, R_) - 1) * pow(n, -1, R_) % R_
    for i in (0, 1, 2, 12345, n - 1):
        wi = pow(w, i, R_)
        li = zn * wi % R_ * pow(TAU - wi, -1, R_) % R_
        assert pyref.point_from_wire(L[i]) == pyref.ec_mul(li, (1, 2)), i
    # cached Lagrange basis: same commitments as the IFFT path and the oracle's literal form
    rnd = random.Random(77)
    for n, srs, wire in ((256, ref_srs, test_srs_wire), (2048, ref_srs, test_srs_wire), (1 << 14, tau_srs, None), (1 << 15, tau_srs, None)):   # 2^15: the Lagrange basis gets per-bit tables (NAF mode)
        poly = k.PolynomialEvalForm(pyref.frs_to_mont([rnd.randrange(R_) for _ in range(n)]))
        via_ifft = kzg.commit_eval_form(poly, srs)
        srs.cache_lagrange(n)
        try:
            via_cache = kzg.commit_eval_form(poly, srs)
            other = kzg.commit_eval_form(k.PolynomialEvalForm(poly.evaluations()[:n // 2]), srs)     # another length: IFFT path
        finally:
            srs.drop_lagrange()
        assert np.array_equal(via_cache, via_ifft), n
        lag = srs.lagrange(n)
        assert np.array_equal(msm_srs(k, lag, poly.evaluations()), via_ifft)
        assert np.array_equal(lag.g1, kzg.g1_ifft(n, srs))
        lag.close()
        if wire is not None and n <= 256:
            rc, lit = orc.commit_eval_form(wire, poly.evaluations(), literal=True)
            assert rc == 0 and np.array_equal(via_cache, lit)
        assert other.shape == (8,)
    with pytest.raises(k.errors.GenericError):
        ref_srs.cache_lagrange(48)


@pytest.mark.parametrize("log_n", [6, 7, 8, 9, 10, 11])
def test_g1_ifft_through_the_per_bit_tables(k, tau_srs, log_n):
    """g1_ifft of 64 .. 2048 points of an SRS that carries per-bit tables (>= 2^15 points): up to 256 points the whole transform as sums
    of table points (k_g1fft_bits: NAF digit lists of the n scalars w^-e / n); 512 / 1024 / 2048 (round 4) with the FIRST STAGE as sums of
    table points and one or two direct stages on lane quads behind it.  Outputs against the known-tau value L_i = l_i(tau) G by
    big-integer arithmetic, and the same transform through the staged kernels (a copy of the first n points as an SRS of its own,
    which is too small for per-bit tables)."""
    kzg = k.KZG.new()
    n = 1 << log_n
    L = kzg.g1_ifft(n, tau_srs)
    w = pyref.root_of_unity(log_n)
    zn = (pow(TAU, n, R_) - 1) * pow(n, -1, R_) % R_
    step = 1 if n <= 128 else 7
    for i in list(range(0, n, step)) + [n - 1]:
        wi = pow(w, i, R_)
        li = zn * wi % R_ * pow(TAU - wi, -1, R_) % R_
        assert pyref.point_from_wire(L[i]) == pyref.ec_mul(li, (1, 2)), i
    small = k.SRS(np.ascontiguousarray(tau_srs.g1[:n]), order=n)
    try:
        assert np.array_equal(kzg.g1_ifft(n, small), L)
    finally:
        small.close()


def test_multi_device_handle_three_contexts_on_one_gpu(k, test_srs_wire, tau_srs):
    """kzg_multi_*: the multi-GPU split behind the C-ABI (one context + one host thread per entry), here three contexts on GPU 0:
    uploaded SRS (3000 reference points, uneven shards) against the oracle, generated known-tau SRS against big-integer values."""
    from rust_kzg_bn254_amd.sharding import MultiKzg
    m = MultiKzg([0, 0, 0])
    m.srs_upload(test_srs_wire)
    assert len(m) == 3000
    for n in (1, 999, 1000, 1001, 2048, 3000):
        sc = rand_scalars(n, 40 + n)
        assert np.array_equal(m.commit_coeff_form(sc), orc.msm_pippenger(test_srs_wire[:n], sc)), n
    with pytest.raises(ValueError, match="polynomial length is not correct"):
        m.commit_coeff_form(rand_scalars(3001, 1))
    ev = rand_scalars(2048, 4048)
    rc, want = orc.commit_eval_form(test_srs_wire, ev, literal=False)
    assert rc == 0 and np.array_equal(m.commit_eval_form(ev), want)
    rc, roots = orc.calculate_roots_of_unity(2048 * 32)
    z = pyref.fr_to_mont(987654321)
    rc, wantp, wanty = orc.compute_proof(test_srs_wire, ev, roots, z, literal=False)
    gp, gy = m.compute_proof(ev, z)
    assert rc == 0 and np.array_equal(gp, wantp) and np.array_equal(gy, wanty)
    zon = roots[5]                                                            # on-domain point
    rc, wantp, wanty = orc.compute_proof(test_srs_wire, ev, roots, zon, literal=False)
    gp, gy = m.compute_proof(ev, zon)
    assert np.array_equal(gp, wantp) and np.array_equal(gy, wanty)
    with pytest.raises(ValueError, match="inconsistent length"):
        m.compute_proof(ev, z, n_roots=1024)
    m.close()
    m = MultiKzg([0, 0, 0, 0, 0])
    n = 1 << 16
    m.srs_generate(TAU, n)
    rnd = random.Random(5)
    vals = [rnd.randrange(R_) for _ in range(n)]
    ptau = sum(v * pow(TAU, i, R_) for i, v in enumerate(vals)) % R_
    assert pyref.point_from_wire(m.commit_coeff_form(pyref.frs_to_mont(vals))) == pyref.ec_mul(ptau, (1, 2))
    m.close()


def test_multi_device_stream_of_mixed_lengths_keeps_its_slots(k, test_srs_wire):
    """ADVICE r3 (csrc/multi.hip): a resident stream that mixes polynomials too short to reach a device's shard with full-length ones.
    Slots used to be k % depth of the STEP index; after a skipped step two MSMs in flight on one device could map to the same slot and
    the whole stream failed with INVALID_ARG.  Three contexts on GPU 0, shards [0,1000) [1000,2000) [2000,3000): buffer 1 (600
    coefficients) only reaches device 0, buffer 2 (1500) devices 0 and 1; every commitment against the oracle."""
    from rust_kzg_bn254_amd.sharding import MultiKzg
    m = MultiKzg([0, 0, 0])
    m.srs_upload(test_srs_wire)
    lens = {0: 3000, 1: 600, 2: 1500, 3: 2999, 4: 1}
    sc = {b: rand_scalars(n, 900 + b) for b, n in lens.items()}
    want = {b: orc.msm_pippenger(test_srs_wire[:n], sc[b]) for b, n in lens.items()}
    for b in lens:
        m.scalars_upload(b, sc[b])
    order = [0, 1, 0, 3, 1, 1, 2, 0, 4, 3, 2, 1, 0, 0, 4, 4, 3, 1, 2, 0, 3]
    got = m.commit_resident_stream(order)
    for i, b in enumerate(order):
        assert np.array_equal(got[i], want[b]), (i, b)
    m.close()
