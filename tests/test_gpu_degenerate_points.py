"""-m gpu: KZG::g1_ifft (prover/src/kzg.rs:263-285) on DEGENERATE point sets.  The reference transforms whatever `srs.g1[..n]` holds; an
SRS file is never checked for distinct points, and every exceptional case of the group law then occurs inside the butterflies: P + P
(equal points), P + (-P) (identity results), identity inputs, identity outputs (z = 0 in the batched conversion to affine).  All points are
small multiples s_j G of the generator, so L_i = (n^-1 sum_j w^(-ij) s_j) G is known by big-integer arithmetic, and the oracle's literal
restatement is compared as well.

Two carriers: an SRS of exactly n points (no per-bit tables: staged kernels, lane pairs / quads) and an SRS of 2^15 points whose first
2 048 are the degenerate ones (per-bit tables: sums of table points for n <= 256, the table first stage for 512 .. 2 048 -- tables of
the identity are all-identity rows, tables of equal points are equal rows)."""
import hashlib
import random

import numpy as np
import pytest

import oracle as orc
import pyref
from pyref import R_

pytestmark = pytest.mark.gpu
TAU = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % R_
G = (1, 2)


@pytest.fixture(scope="module")
def k():
    import rust_kzg_bn254_amd as k
    k.load()
    k.default_context()
    return k


def patterns(n, rnd):
    small = [0, 1, R_ - 1, 2, R_ - 2, 5, 1, 0]
    return {
        "all equal": [1] * n,                                               # every butterfly adds P + P or P - P
        "alternating +-G": [1 if j % 2 == 0 else R_ - 1 for j in range(n)],
        "one point, the rest identity": [1] + [0] * (n - 1),
        "identity everywhere": [0] * n,
        "second half identity": [3] * (n // 2) + [0] * (n - n // 2),
        "few small multiples": [rnd.choice(small) for _ in range(n)],
    }


_MULT = {}


def mult(s):
    s %= R_
    if s not in _MULT:
        _MULT[s] = pyref.ec_mul(s, G)
    return _MULT[s]


def expected_closed_form(name, n):
    """L as scalars of G for the patterns whose inverse transform is known in closed form (None: use the O(n^2) definition)."""
    if name == "all equal":
        return [1] + [0] * (n - 1)
    if name == "alternating +-G":
        return [1 if i == n // 2 else 0 for i in range(n)] if n > 1 else [1]
    if name == "one point, the rest identity":
        return [pow(n, -1, R_)] * n
    if name == "identity everywhere":
        return [0] * n
    return None


@pytest.mark.parametrize("n", [2, 8, 64, 256, 512, 1024, 2048])
def test_g1_ifft_of_degenerate_point_sets(k, n):
    rnd = random.Random(9000 + n)
    kzg = k.KZG.new()
    big_pts = None
    for name, s in patterns(n, rnd).items():
        pts = pyref.points_to_wire([mult(v) for v in s])
        rc, want = orc.g1_ifft(pts, n)
        assert rc == 0
        coef = expected_closed_form(name, n)
        if coef is None and n <= 256:
            coef = pyref.dft(s, inverse=True)
        if coef is not None:                                            # big integers pin the oracle on these inputs too
            for i in ([0, 1, n // 2, n - 1] if n > 64 else range(n)):
                assert pyref.point_from_wire(want[i]) == pyref.ec_mul(coef[i], G), (name, n, i)
        srs = k.SRS(pts)
        try:
            got = kzg.g1_ifft(n, srs)
        finally:
            srs.close()
        assert np.array_equal(got, want), ("SRS of n points", name, n)
        if n >= 64:
            # the same points at the head of a 2^15-point SRS with per-bit tables
            if big_pts is None:
                tail = k.SRS.generate(TAU, 1 << 15)
                big_pts = tail.g1
                tail.close()
            full = big_pts.copy()
            full[:n] = pts
            big = k.SRS(full)
            try:
                assert k._lib.load().kzg_srs_has_bit_tables(big.handle, 1) == 1
                got2 = kzg.g1_ifft(n, big)
            finally:
                big.close()
            assert np.array_equal(got2, want), ("head of a 2^15-point SRS", name, n)


def test_commitments_and_lagrange_cache_over_degenerate_points(k):
    """commit_eval_form over such an SRS (IFFT of the scalars + MSM, and the cached Lagrange basis: an SRS whose points are mostly the
    identity) == the big-integer value."""
    n = 256
    rnd = random.Random(5)
    s = [rnd.choice([0, 1, R_ - 1, 2, 7]) for _ in range(n)]
    pts = pyref.points_to_wire([mult(v) for v in s])
    evals = [rnd.randrange(R_) for _ in range(n)]
    coeffs = pyref.dft(evals, inverse=True)
    want = pyref.ec_mul(sum(c * v for c, v in zip(coeffs, s)) % R_, G)
    kzg = k.KZG.new()
    srs = k.SRS(pts)
    try:
        poly = k.PolynomialEvalForm(pyref.frs_to_mont(evals))
        assert pyref.point_from_wire(kzg.commit_eval_form(poly, srs)) == want
        srs.cache_lagrange(n)
        try:
            assert pyref.point_from_wire(kzg.commit_eval_form(poly, srs)) == want
        finally:
            srs.drop_lagrange()
        for name in ("all equal", "one point, the rest identity", "identity everywhere"):
            s2 = patterns(n, rnd)[name]
            srs2 = k.SRS(pyref.points_to_wire([mult(v) for v in s2]))
            try:
                w2 = pyref.ec_mul(sum(c * v for c, v in zip(coeffs, s2)) % R_, G)
                assert pyref.point_from_wire(kzg.commit_eval_form(poly, srs2)) == w2, name
                srs2.cache_lagrange(n)
                assert pyref.point_from_wire(kzg.commit_eval_form(poly, srs2)) == w2, name + " (cached Lagrange basis)"
            finally:
                srs2.close()
    finally:
        srs.close()


@pytest.mark.parametrize("log_srs", [12, 15])
def test_srs_msm_over_degenerate_points(k, log_srs):
    """The SRS path of the MSM (window tables; per-bit tables + NAF digits at 2^15 points; sums of table points up to 4 096 pairs; the fused
    sparse mode) over an UPLOADED point set with equal points, +-P pairs and identities: the table builds double the identity and equal
    rows, the buckets see P + P and P - P (SURVEY 8d edge sets, which tests/test_gpu_parity.py runs on caller bases only).  Expected value:
    (sum_j c_j s_j) G by big integers."""
    import ctypes as C
    N = 1 << log_srs
    rnd = random.Random(700 + log_srs)
    ctx = k.default_context(); lib = k._lib.load()
    small = [0, 1, R_ - 1, 2, R_ - 2, 5, 1, 1]
    for name, s in (("all equal", [1] * N), ("alternating +-G", [1 if j % 2 == 0 else R_ - 1 for j in range(N)]),
                    ("mostly identity", [rnd.choice([0, 0, 0, 1]) for _ in range(N)]), ("few small multiples", [rnd.choice(small) for _ in range(N)])):
        srs = k.SRS(pyref.points_to_wire([mult(v) for v in s]))
        try:
            for n in sorted({1, 2, 100, 4096, 4097, 8192, N // 2 + 1, N}):
                if n > N:
                    continue
                for kind in ("uniform", "equal", "pairs cancel"):
                    if kind == "uniform":
                        c = [rnd.randrange(R_) for _ in range(n)]
                    elif kind == "equal":
                        c = [rnd.randrange(R_)] * n                         # equal scalars on equal points: one bucket per digit
                    else:
                        half = [rnd.randrange(R_) for _ in range((n + 1) // 2)]
                        c = [half[j // 2] for j in range(n)]                 # c_2j = c_2j+1: cancels on the alternating set
                    out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
                    assert lib.kzg_msm_g1_srs(ctx.handle, srs.handle, 0, k._lib.ptr(pyref.frs_to_mont(c)), n, k._lib.ptr(out), C.byref(inf)) == 0
                    want = pyref.ec_mul(sum(a * b for a, b in zip(c, s)) % R_, G)
                    assert pyref.point_from_wire(out) == want, (log_srs, name, n, kind)
                    assert bool(inf.value) == (want is None)
            # an offset window and the asynchronous form
            off, n = 3, min(N - 3, 5000)
            c = [rnd.randrange(R_) for _ in range(n)]
            out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
            assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, off, k._lib.ptr(pyref.frs_to_mont(c)), n, 1) == 0
            assert lib.kzg_msm_g1_srs_end(ctx.handle, 1, k._lib.ptr(out), C.byref(inf), None) == 0
            assert pyref.point_from_wire(out) == pyref.ec_mul(sum(a * b for a, b in zip(c, s[off:])) % R_, G), (log_srs, name, "offset")
        finally:
            srs.close()
