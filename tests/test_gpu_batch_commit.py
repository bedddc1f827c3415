"""-m gpu: batched commitments (kzg_commit_*_form_batch: many polynomials of one length against one SRS in ONE kernel sequence --
width-8 NAF digits over the SRS's per-bit tables, 64 buckets per polynomial).  No counterpart in the reference, which commits one
polynomial per call (prover/src/kzg.rs:84-125): the batch must equal that many single commitments, the oracle and known-tau values."""
import ctypes as C
import os
import random

import numpy as np
import pytest

import oracle as orc
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
def tau_srs(k):
    s = k.SRS.generate(TAU, 1 << 15)
    yield s
    s.close()


def rand_poly(k, n, seed, mod=R_):
    rnd = random.Random(seed)
    return k.PolynomialCoeffForm(pyref.frs_to_mont([rnd.randrange(mod) for _ in range(n)]))


def test_batch_equals_singles_and_oracle_on_the_reference_srs(k, test_srs_wire):
    """3000 reference points (no per-bit tables until the first batched call builds them): 5 polynomials of 512 coefficients."""
    srs = k.SRS(test_srs_wire, order=3000)
    kzg = k.KZG.new()
    polys = [rand_poly(k, 512, 100 + i) for i in range(5)]
    got = kzg.commit_coeff_form_batch(polys, srs)
    assert got.shape == (5, 8)
    for p, g in zip(polys, got):
        assert np.array_equal(g, kzg.commit_coeff_form(p, srs))
        assert np.array_equal(g, orc.msm_pippenger(test_srs_wire[:512], p.coeffs()))
    # a second call (tables already there), another length, a single polynomial
    one = kzg.commit_coeff_form_batch([rand_poly(k, 1000, 7)], srs)
    assert np.array_equal(one[0], kzg.commit_coeff_form(rand_poly(k, 1000, 7), srs))
    assert kzg.commit_coeff_form_batch([], srs).shape == (0, 8)
    with pytest.raises(k.errors.SerializationError):
        kzg.commit_coeff_form_batch([rand_poly(k, 4096, 1)], srs)
    with pytest.raises(k.errors.GenericError):
        kzg.commit_coeff_form_batch([rand_poly(k, 512, 1), rand_poly(k, 256, 2)], srs)
    srs.close()


def test_batch_of_special_polynomials_known_tau(k, tau_srs):
    """1 030 polynomials of 512 coefficients (two launches: 1 024 + 6) on a known-tau SRS: random ones, the zero polynomial, one-hot,
    all r - 1, all 1, 2^253 - 1 beside 2^36 - 1 (the digit patterns that exposed the recoder's miscompiled branch): every commitment
    against (sum_i c_i tau^i) G by big-integer arithmetic."""
    kzg = k.KZG.new()
    n, count = 512, 1030
    rnd = random.Random(9)
    tp = [pow(TAU, i, R_) for i in range(n)]
    rows = []
    for j in range(count):
        kind = j % 8
        if kind == 0:
            v = [0] * n
        elif kind == 1:
            v = [0] * n; v[(j * 37) % n] = rnd.randrange(R_)
        elif kind == 2:
            v = [R_ - 1] * n
        elif kind == 3:
            v = [1] * n
        elif kind == 4:
            v = [(1 << 253) - 1 if i % 2 == 0 else (1 << 36) - 1 for i in range(n)]
        else:
            v = [rnd.randrange(R_) for _ in range(n)]
        rows.append(v)
    polys = [k.PolynomialCoeffForm(pyref.frs_to_mont(v)) for v in rows]
    got = kzg.commit_coeff_form_batch(polys, tau_srs)
    for j in list(range(0, 40)) + list(range(1000, count)):
        s = sum(c * t for c, t in zip(rows[j], tp)) % R_
        want = pyref.ec_mul(s, (1, 2)) if s else None
        assert pyref.point_from_wire(got[j]) == want, j
    # and all of them against single commitments at a stride
    for j in range(0, count, 97):
        assert np.array_equal(got[j], kzg.commit_coeff_form(polys[j], tau_srs)), j


def test_batch_2048_coefficients_equals_singles(k, tau_srs):
    kzg = k.KZG.new()
    polys = [rand_poly(k, 2048, 500 + i, mod=1 << 248) for i in range(64)]
    got = kzg.commit_coeff_form_batch(polys, tau_srs)
    for p, g in zip(polys, got):
        assert np.array_equal(g, kzg.commit_coeff_form(p, tau_srs))


@pytest.mark.parametrize("log_n,count", [(13, 19), (14, 5), (15, 6)])
def test_batch_of_long_polynomials_equals_singles(k, tau_srs, log_n, count):
    """From 2^13 coefficients a polynomial gets whole units of 4 096 buckets (13 bucket bits: 16 polynomials per launch; 15: four) and the
    second reduction level + host epilogue of a single MSM, per polynomial: batch == single calls, incl. a zero polynomial and r - 1."""
    kzg = k.KZG.new()
    n = 1 << log_n
    polys = [rand_poly(k, n, 900 + 31 * log_n + i, mod=(1 << 248) if i % 2 else R_) for i in range(count - 2)]
    polys.append(k.PolynomialCoeffForm(pyref.frs_to_mont([0] * n)))
    polys.append(k.PolynomialCoeffForm(pyref.frs_to_mont([R_ - 1] * n)))
    got = kzg.commit_coeff_form_batch(polys, tau_srs)
    for j, (p, g) in enumerate(zip(polys, got)):
        assert np.array_equal(g, kzg.commit_coeff_form(p, tau_srs)), (log_n, j)
    assert not got[count - 2].any()


def test_eval_form_and_blob_batches(k, tau_srs):
    """commit_eval_form_batch == commit_eval_form per polynomial (the batch goes through the cached Lagrange basis: the reference's
    literal form, kzg.rs:98-100); blobs of several lengths; error cases of the single call."""
    kzg = k.KZG.new()
    rnd = random.Random(3)
    polys = [k.PolynomialEvalForm(pyref.frs_to_mont([rnd.randrange(R_) for _ in range(1024)])) for _ in range(8)]
    got = kzg.commit_eval_form_batch(polys, tau_srs)
    for p, g in zip(polys, got):
        assert np.array_equal(g, kzg.commit_eval_form(p, tau_srs))
    for n in (1, 2, 64):                                   # tiny Lagrange bases (no window tables of their own: per-bit tables built on demand)
        tiny = [k.PolynomialEvalForm(pyref.frs_to_mont([rnd.randrange(R_) for _ in range(n)])) for _ in range(3)]
        for p, g in zip(tiny, kzg.commit_eval_form_batch(tiny, tau_srs)):
            assert np.array_equal(g, kzg.commit_eval_form(p, tau_srs)), n
    blobs = [k.Blob.from_raw_data(bytes(rnd.randrange(32, 127) for _ in range(ln))) for ln in (1000, 31 * 64, 5000, 31 * 64 - 3, 20000, 999)]
    gotb = kzg.commit_blob_batch(blobs, tau_srs)
    for b, g in zip(blobs, gotb):
        assert np.array_equal(g, kzg.commit_blob(b, tau_srs))
    lib = k._lib.load()
    ctx = tau_srs.ctx
    out = np.zeros((2, 8), np.uint64)
    data = np.zeros((2 * 48, 4), np.uint64)
    assert lib.kzg_commit_eval_form_batch(ctx.handle, tau_srs.handle, k._lib.ptr(data), 48, 2, k._lib.ptr(out), None) == k._lib.ERR_NOT_POWER_OF_TWO
    assert lib.kzg_commit_eval_form_batch(ctx.handle, tau_srs.handle, k._lib.ptr(data), 1 << 16, 2, k._lib.ptr(out), None) == k._lib.ERR_SRS_CAPACITY_EXCEEDED
    assert lib.kzg_commit_coeff_form_batch(ctx.handle, tau_srs.handle, k._lib.ptr(data), 1 << 16, 2, k._lib.ptr(out), None) == k._lib.ERR_POLY_LENGTH
    tau_srs.drop_lagrange()


def test_grouped_stream_of_shard_sized_commitments(k, tau_srs):
    """ShardedMsm.commit_stream with several steps per launch (kzg_msm_g1_srs_device_begin_batch / _end_batch: scalar sets in SEPARATE
    device buffers, one batched launch): the same commitments as one launch per step, at group sizes that do and do not divide the
    stream; the XYZZ partials of a group fold to the same points (the world > 1 path); capacity and slot misuse are errors."""
    import torch
    from rust_kzg_bn254_amd.sharding import ShardedMsm, fold_partials
    n = 1 << 15
    rnd = random.Random(11)
    bufs = []
    for j in range(5):
        vals = [rnd.randrange(R_) for _ in range(64)]
        arr = np.ascontiguousarray(np.tile(pyref.frs_to_mont(vals), (n // 64, 1)))          # 2^15 scalars with period 64
        arr[j] = pyref.frs_to_mont([j])[0]
        bufs.append(torch.from_numpy(arr.view(np.int64)).cuda())
    torch.cuda.synchronize()
    sh = ShardedMsm(tau_srs.ctx, n)
    ptrs = [bufs[j % 5].data_ptr() for j in range(11)]
    want = list(sh.commit_stream(tau_srs, ptrs, depth=2, group=1))
    assert len(want) == 11 and not np.array_equal(want[0], want[1])
    for depth, group in ((3, 4), (2, 3), (1, 2), (3, None)):            # None: auto_group()
        got = list(sh.commit_stream(tau_srs, ptrs, depth=depth, group=group))
        assert len(got) == 11 and all(np.array_equal(a, b) for a, b in zip(got, want)), (depth, group)
    assert sh.auto_group() == 4
    os.environ["KZG_SHARD_GROUP_AUTO"] = "0"
    try:
        assert sh.auto_group() == 1
    finally:
        del os.environ["KZG_SHARD_GROUP_AUTO"]
    sh.begin_group(tau_srs, ptrs[:3], 1)
    parts = sh._end_partials(1, 3)
    for j in range(3):
        assert np.array_equal(fold_partials(parts[j].reshape(1, 16)), want[j])
    lib = k._lib.load()
    assert int(lib.kzg_msm_batch_capacity(n)) == 4 and int(lib.kzg_msm_batch_capacity(512)) == 1024 and int(lib.kzg_msm_batch_capacity(1 << 18)) == 2
    with pytest.raises(ValueError):
        sh.begin_group(tau_srs, ptrs[:5], 0)                       # more than one launch takes
    sh.begin_group(tau_srs, ptrs[:2], 0)
    one = np.zeros(8, np.uint64); inf = C.c_uint8(0)
    assert lib.kzg_msm_g1_srs_end(tau_srs.ctx.handle, 0, k._lib.ptr(one), C.byref(inf), None) == k._lib.ERR_INVALID_ARG     # a batched launch is collected by _end_batch
    assert all(np.array_equal(a, b) for a, b in zip(sh._end_group(0, 2), want[:2]))


def test_naf_mode_on_ragged_sizes_and_slices(k):
    """The NAF mode (per-bit tables) on an SRS and MSM lengths that are not powers of two, and on slices of the SRS (offset != 0: the tables
    are srs.n points apart, the slice starts inside every one of them): against sum_i c_i tau^(offset + i) on a known-tau SRS."""
    n_srs = 40000
    srs = k.SRS.generate(TAU, n_srs)
    lib = k._lib.load()
    rnd = random.Random(21)
    tp = [1]
    for _ in range(n_srs - 1):
        tp.append(tp[-1] * TAU % R_)
    try:
        for offset, n in ((0, 40000), (0, 33001), (5000, 30000), (7, 16384), (39999 - 16500, 16500)):
            vals = [rnd.randrange(R_) if i % 3 else rnd.randrange(1 << 248) for i in range(n)]
            vals[0] = 0; vals[n - 1] = R_ - 1
            sc = np.ascontiguousarray(pyref.frs_to_mont(vals), dtype=np.uint64).reshape(-1, 4)
            out = np.zeros(8, np.uint64); inf = C.c_uint8(7)
            rc = lib.kzg_msm_g1_srs(srs.ctx.handle, srs.handle, offset, k._lib.ptr(sc), n, k._lib.ptr(out), C.byref(inf))
            assert rc == 0
            s = sum(v * tp[offset + i] for i, v in enumerate(vals)) % R_
            assert pyref.point_from_wire(out) == pyref.ec_mul(s, (1, 2)), (offset, n)
    finally:
        srs.close()


def test_sparse_msm_with_heavy_buckets(k, test_srs_wire):
    """The fused first reduction level (sparse MSMs: at most 2.5 entries per bucket on average) with HEAVY buckets: blob-like scalars
    (< 2^248: the short top window of a small SRS's 13-bit tables holds one of two digits) and vectors of one repeated scalar (n
    entries in each of ~20 buckets) on 512 / 700 / 1024 points of the reference SRS, against the oracle."""
    lib = k._lib.load()
    rnd = random.Random(31)
    for n_srs in (512, 1024):
        srs = k.SRS(test_srs_wire[:n_srs], order=n_srs)
        try:
            for n, kind in ((n_srs, "blob"), (n_srs, "same"), (min(700, n_srs), "blob"), (n_srs, "two")):
                if kind == "blob":
                    vals = [int.from_bytes(bytes([0] + [rnd.randrange(32, 127) for _ in range(31)]), "big") for _ in range(n)]
                elif kind == "same":
                    vals = [rnd.randrange(R_)] * n
                else:
                    a, b = rnd.randrange(R_), rnd.randrange(1 << 100)
                    vals = [a if i % 3 else b for i in range(n)]
                sc = np.ascontiguousarray(pyref.frs_to_mont(vals), dtype=np.uint64).reshape(-1, 4)
                out = np.zeros(8, np.uint64); inf = C.c_uint8(7)
                assert lib.kzg_msm_g1_srs(srs.ctx.handle, srs.handle, 0, k._lib.ptr(sc), n, k._lib.ptr(out), C.byref(inf)) == 0
                assert np.array_equal(out, orc.msm_pippenger(test_srs_wire[:n], sc)), (n_srs, n, kind)
        finally:
            srs.close()
