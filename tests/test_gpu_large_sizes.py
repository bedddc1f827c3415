"""-m gpu: sizes beyond the tables.  The reference accepts polynomials of up to 2^28 elements
(primitives/src/polynomial.rs:42, :170) and SRS files of up to 2^28 points (primitives/src/consts.rs:66); the GPU path had been
oracle-checked to 2^24 (NTT) and a 2^21-point SRS (MSM).

* Fr NTT at 2^25 and 2^26 (1 / 2 GiB of data; three passes, the lookup-twiddle path): forward transform bit for bit against the
  oracle, exact round trip, the inverse against the oracle at 2^25, and the definition F[i] = sum_j a_j w^(ij) at sparse indices.
* commitments over a 2^23-point SRS (window tables, NO per-bit tables: the table-less side of srs_build_bit_tables' 2^22 limit) and a
  2^24-point SRS (16 launches of 2^20 pairs: the cap of the asynchronous calls), checked against (sum_i c_i tau^i mod r) G1 by big
  integers -- never against another run of the HIP path.  Step times are printed (pytest -s) and reported by bench.py `secondary`.
Host memory: ~7 GiB at the peak (2^26 elements = 2 GiB per array)."""
import ctypes as C
import hashlib
import time

import numpy as np
import pytest

import oracle as orc
import pyref
from pyref import R_

pytestmark = pytest.mark.gpu
TAU = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % R_
MONT = (1 << 256) % R_


@pytest.fixture(scope="module")
def k():
    import rust_kzg_bn254_amd as k
    k.load()
    k.default_context()
    return k


def random_canonical(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return a


@pytest.mark.parametrize("log_n", [25, 26])
def test_ntt_2_25_and_2_26_match_the_oracle(k, log_n):
    """2^25: the whole forward transform against the oracle, bit for bit.  2^26 (three passes, like 2^25): the decimation-in-time identity
    F[i] = E[i mod n/2] + w^i O[i mod n/2] with E / O the 2^25-point transforms of the even / odd elements -- the size the oracle pins --
    at 256 random i plus the corners, by big integers; then for both sizes the exact round trip and the definition on a sparse input."""
    n = 1 << log_n
    a = random_canonical(n, 9100 + log_n)
    ctx = k.default_context(); lib = k._lib.load()
    f = a.copy()
    t0 = time.perf_counter()
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(f), n, 0) == 0
    t_gpu = time.perf_counter() - t0
    if log_n == 25:
        t0 = time.perf_counter()
        want = orc.fr_ntt_mt(a, inverse=False)
        t_cpu = time.perf_counter() - t0
        print("2^%d forward NTT: GPU incl. PCIe both ways %.3f s, oracle (%d threads) %.1f s" % (log_n, t_gpu, orc.host_cpus(), t_cpu))
        assert np.array_equal(f, want), log_n
        del want
    else:
        h = n // 2
        ev = np.ascontiguousarray(a[0::2]); od = np.ascontiguousarray(a[1::2])
        assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(ev), h, 0) == 0 and lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(od), h, 0) == 0
        w = pyref.root_of_unity(log_n)
        idx = [0, 1, h - 1, h, h + 1, n - 1] + [int(v) for v in np.random.default_rng(26).integers(0, n, size=256)]
        for i in idx:
            want_i = (pyref.fr_from_mont(ev[i % h]) + pow(w, i, R_) * pyref.fr_from_mont(od[i % h])) % R_
            assert pyref.fr_from_mont(f[i]) == want_i, i
        del ev, od
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(f), n, 1) == 0            # round trip (pins the inverse: it undoes an oracle-exact forward transform)
    assert np.array_equal(f, a)
    del f
    a[:] = 0
    idxs, vals = [0, 1, 5_000_001, n - 1], [3, 5, 7, 11]
    for j, v in zip(idxs, vals):
        a[j] = pyref.fr_to_mont(v)
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(a), n, 0) == 0
    w = pyref.root_of_unity(log_n)
    for i in (0, 1, 2, 1234567, n // 2, n // 2 + 1, n - 1):
        assert pyref.fr_from_mont(a[i]) == sum(v * pow(w, i * j, R_) for j, v in zip(idxs, vals)) % R_, (log_n, i)


def _blob_like(n, seed):
    """canonical values < 2^248 (31 payload bytes behind a zero byte, helpers.rs:823-840) as python ints + wire array"""
    rng = np.random.default_rng(seed)
    raw = rng.integers(32, 127, size=(n, 31), dtype=np.uint8)
    be = np.zeros((n, 32), np.uint8); be[:, 1:] = raw
    vals = [int.from_bytes(be[i].tobytes(), "big") for i in range(n)]
    return vals


def _to_wire(vals):
    return np.frombuffer(b"".join((v * MONT % R_).to_bytes(32, "little") for v in vals), dtype=np.uint64).reshape(-1, 4).copy()


def _expected(vals, offset=0):
    acc, tp = 0, pow(TAU, offset, R_)
    for v in vals:
        acc = (acc + v * tp) % R_
        tp = tp * TAU % R_
    return pyref.ec_mul(acc, (1, 2))


@pytest.mark.parametrize("log_srs", [23, 24])
def test_commitments_over_2_23_and_2_24_point_srs(k, log_srs):
    ctx = k.default_context(); lib = k._lib.load()
    n_srs = 1 << log_srs
    t0 = time.perf_counter()
    srs = k.SRS.generate(TAU, n_srs)
    print("SRS 2^%d: generated + tables in %.2f s; per-bit tables: %d" % (log_srs, time.perf_counter() - t0, lib.kzg_srs_has_bit_tables(srs.handle, 0)))
    try:
        assert lib.kzg_srs_has_bit_tables(srs.handle, 0) == 0          # above 2^22 points: window tables only
        out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
        # the whole SRS; a ragged length that ends inside a launch; one 2^20 launch at an offset straddling launch boundaries; a small MSM at the far end
        cases = [(0, n_srs), (0, 2 * (1 << 20) + 4321), ((3 << 20) - 1000, 1 << 20), (n_srs - 3000, 2048)]
        whole = None
        for ci, (offset, n) in enumerate(cases):
            if ci == 0:
                # the whole SRS: 4 096 full-width constants repeated with period 4 096 (every point and every launch takes part; the expected value
                # has the closed form of the 2^25 / 2^26 test below, so no 2^24 big-integer products on the host)
                period = 4096
                rng = np.random.default_rng(log_srs)
                ks = [int.from_bytes(rng.bytes(40), "little") % R_ for _ in range(period)]
                wire = np.ascontiguousarray(np.tile(_to_wire(ks), (n // period, 1)))
                acc, tp = 0, 1
                for v in ks:
                    acc = (acc + v * tp) % R_
                    tp = tp * TAU % R_
                geo = lambda cnt: (pow(TAU, cnt, R_) - 1) * pow(pow(TAU, period, R_) - 1, -1, R_) % R_          # noqa: E731
                want = pyref.ec_mul(acc * geo(n) % R_, (1, 2))
                whole = (wire, want, pyref.ec_mul(acc * geo(1 << 20) % R_, (1, 2)))
            else:                                                            # full-width blob-like scalars, big-integer expectation
                vals = _blob_like(n, 50 + ci + log_srs)
                wire = _to_wire(vals)
                want = _expected(vals, offset)
            t0 = time.perf_counter()
            assert lib.kzg_msm_g1_srs(ctx.handle, srs.handle, offset, k._lib.ptr(wire), n, k._lib.ptr(out), C.byref(inf)) == 0
            dt = time.perf_counter() - t0
            print("  SRS 2^%d, MSM of %d pairs at offset %d from host scalars: %.2f ms" % (log_srs, n, offset, dt * 1e3))
            assert pyref.point_from_wire(out) == want, (log_srs, offset, n)
        # asynchronous form on the whole SRS (16 launches at 2^24: the documented cap) beside a second slot
        wire, want, want_first = whole
        assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, 0, k._lib.ptr(wire), n_srs, 0) == 0
        assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, 0, k._lib.ptr(wire), 1 << 20, 1) == 0
        o2 = np.zeros(8, np.uint64)
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, k._lib.ptr(out), C.byref(inf), None) == 0
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 1, k._lib.ptr(o2), C.byref(inf), None) == 0
        assert pyref.point_from_wire(out) == want and pyref.point_from_wire(o2) == want_first
    finally:
        srs.close()


@pytest.mark.parametrize("log_srs", [25, 26])
def test_commitment_over_2_25_and_2_26_point_srs_periodic_scalars(k, log_srs):
    """A commitment over EVERY point of a 2^25-point SRS (window tables, 32 launches of 2^20 pairs) and of a 2^26-point SRS (its window
    tables would take 64 GiB: above the 48 GiB cap of srs_precompute, so this is the table-less generic mode: 19 windows of 2^13
    buckets, Horner on the host).  Scalars: 4 096 full-width random constants repeated with period 4 096, so that the expected value
    has a closed form -- sum_m k_m tau^m (tau^(4096 Q) - 1) / (tau^4096 - 1) -- and 2^26 big-integer products are not needed on the host;
    4 096 x 15 distinct (window, digit) pairs keep the buckets evenly filled."""
    ctx = k.default_context(); lib = k._lib.load()
    n = 1 << log_srs
    period = 4096
    rng = np.random.default_rng(100 + log_srs)
    ks = [int.from_bytes(rng.bytes(40), "little") % R_ for _ in range(period)]
    block = _to_wire(ks)
    wire = np.ascontiguousarray(np.tile(block, (n // period, 1)))
    q = n // period
    geo = (pow(TAU, period * q, R_) - 1) * pow(pow(TAU, period, R_) - 1, -1, R_) % R_
    acc, tp = 0, 1
    for v in ks:
        acc = (acc + v * tp) % R_
        tp = tp * TAU % R_
    want = pyref.ec_mul(acc * geo % R_, (1, 2))
    t0 = time.perf_counter()
    srs = k.SRS.generate(TAU, n)
    t_srs = time.perf_counter() - t0
    try:
        out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
        t0 = time.perf_counter()
        assert lib.kzg_msm_g1_srs(ctx.handle, srs.handle, 0, k._lib.ptr(wire), n, k._lib.ptr(out), C.byref(inf)) == 0
        dt = time.perf_counter() - t0
        print("SRS 2^%d (set-up %.1f s): commitment over all %d points from host scalars: %.1f ms (%.2f ms per 2^20 pairs)" % (log_srs, t_srs, n, dt * 1e3, dt * 1e3 / (n >> 20)))
        assert pyref.point_from_wire(out) == want, log_srs
        # the asynchronous form reaches 2^26 pairs since round 4 (64 launches of 2^20 over the 2^25-point tables, four of 2^24 in generic mode);
        # beside it a 2^24-pair MSM at an offset on a second slot
        assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, 0, k._lib.ptr(wire), n, 0) == 0
        assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, 5, k._lib.ptr(wire), 1 << 24, 1) == 0
        o2 = np.zeros(8, np.uint64)
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, k._lib.ptr(out), C.byref(inf), None) == 0
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 1, k._lib.ptr(o2), C.byref(inf), None) == 0
        assert pyref.point_from_wire(out) == want
        geo24 = (pow(TAU, 1 << 24, R_) - 1) * pow(pow(TAU, period, R_) - 1, -1, R_) % R_
        assert pyref.point_from_wire(o2) == pyref.ec_mul(acc * geo24 % R_ * pow(TAU, 5, R_) % R_, (1, 2))
        assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, 0, k._lib.ptr(wire), n, 7) != 0            # not a slot
    finally:
        srs.close()


@pytest.mark.parametrize("log_n", [22, 24])
def test_eval_form_commitment_and_proofs_at_2_22_and_2_24(k, log_n):
    """commit_eval_form (kzg.rs:84-104) and compute_proof (kzg.rs:128-178, :237-260), off the domain and on it, at 2^22 and 2^24
    evaluations (the reference accepts 2^28; the proof pipeline had been checked to 2^20).  The polynomial has 2 000 random coefficients at
    random degrees (degree n - 1 among them), so f(tau), f(z) and the quotient's value at tau are cheap big-integer sums; its n evaluations
    -- dense -- come from the GPU NTT, which tests/test_gpu_ntt_sizes.py and the test above pin against the oracle up to 2^26."""
    n = 1 << log_n
    ctx = k.default_context(); lib = k._lib.load()
    rng = np.random.default_rng(2200 + log_n)
    idx = sorted(set([0, 1, n - 1] + [int(v) for v in rng.integers(0, n, size=2000)]))
    cs = [int.from_bytes(rng.bytes(40), "little") % R_ for _ in idx]
    f = lambda x: sum(c * pow(x, j, R_) for c, j in zip(cs, idx)) % R_          # noqa: E731
    coeffs = np.zeros((n, 4), np.uint64)
    coeffs[idx] = pyref.frs_to_mont(cs)
    evals = coeffs.copy()
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(evals), n, 0) == 0
    w = pyref.root_of_unity(log_n)
    for m in (0, 1, 12345, n - 1):
        assert pyref.fr_from_mont(evals[m]) == f(pow(w, m, R_)), m                # the NTT's output is the evaluation vector
    srs = k.SRS.generate(TAU, n)
    try:
        out = np.zeros(8, np.uint64); inf = C.c_uint8(0); y = np.zeros(4, np.uint64)
        ftau = f(TAU)
        t0 = time.perf_counter()
        assert lib.kzg_commit_eval_form(ctx.handle, srs.handle, k._lib.ptr(evals), n, k._lib.ptr(out), C.byref(inf)) == 0
        t_commit = time.perf_counter() - t0
        assert pyref.point_from_wire(out) == pyref.ec_mul(ftau, (1, 2)), log_n
        assert lib.kzg_commit_coeff_form(ctx.handle, srs.handle, k._lib.ptr(coeffs), n, k._lib.ptr(out), C.byref(inf)) == 0
        assert pyref.point_from_wire(out) == pyref.ec_mul(ftau, (1, 2)), log_n
        z_off = int.from_bytes(rng.bytes(40), "little") % R_
        for z, which in ((z_off, "off the domain"), (pow(w, 3 * n // 7, R_), "on the domain"), (1, "z = 1")):
            zw = pyref.fr_to_mont(z)
            t0 = time.perf_counter()
            assert lib.kzg_compute_proof(ctx.handle, srs.handle, k._lib.ptr(evals), n, None, n, k._lib.ptr(zw), k._lib.ptr(out), C.byref(inf), k._lib.ptr(y)) == 0
            dt = time.perf_counter() - t0
            fz = f(z)
            assert pyref.fr_from_mont(y) == fz, (log_n, which)
            assert pyref.point_from_wire(out) == pyref.ec_mul((ftau - fz) * pow(TAU - z, -1, R_) % R_, (1, 2)), (log_n, which)
            print("  2^%d evaluations: commit_eval_form %.1f ms, compute_proof %s %.1f ms (host buffers)" % (log_n, t_commit * 1e3, which, dt * 1e3))
    finally:
        srs.close()


def test_naf_bucket_bits_follow_the_in_flight_state(k):
    """An MSM of 2^18 .. 2^19 - 1 pairs over the per-bit tables runs with 2^14 buckets when nothing else is in flight on its context and
    with 2^15 when another MSM is (msm.hip make_plan; the reference's bench_kzg_commit_8mb shape is the lone case).  Both plans, at the
    edges of the range and inside it, against sum_i c_i tau^i G1 by big integers."""
    ctx = k.default_context(); lib = k._lib.load()
    N = 1 << 19
    srs = k.SRS.generate(TAU, N)
    rng = np.random.default_rng(1819)
    try:
        other = _to_wire([int.from_bytes(rng.bytes(40), "little") % R_ for _ in range(1 << 14)])
        acc, tp = 0, 1
        for v in pyref.frs_from_mont(other):
            acc = (acc + v * tp) % R_; tp = tp * TAU % R_
        want_other = pyref.ec_mul(acc, (1, 2))
        all_vals = [int.from_bytes(rng.bytes(40), "little") % R_ for _ in range(N)]       # every case commits a prefix of the same N scalars
        all_wire = _to_wire(all_vals)
        prefix, acc, tp = {}, 0, 1
        sizes = ((1 << 18) - 1, 1 << 18, (3 << 17) + 5, N - 1, N)
        for i, v in enumerate(all_vals):
            acc = (acc + v * tp) % R_; tp = tp * TAU % R_
            if i + 1 in sizes:
                prefix[i + 1] = acc
        for n in sizes:
            wire = np.ascontiguousarray(all_wire[:n])
            want = pyref.ec_mul(prefix[n], (1, 2))
            out = np.zeros(8, np.uint64); o2 = np.zeros(8, np.uint64); inf = C.c_uint8(0)
            assert lib.kzg_msm_g1_srs(ctx.handle, srs.handle, 0, k._lib.ptr(wire), n, k._lib.ptr(out), C.byref(inf)) == 0           # alone
            assert pyref.point_from_wire(out) == want, (n, "alone")
            assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, 0, k._lib.ptr(other), len(other), 1) == 0                        # another MSM in flight
            assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, 0, k._lib.ptr(wire), n, 0) == 0
            assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, k._lib.ptr(out), C.byref(inf), None) == 0
            assert lib.kzg_msm_g1_srs_end(ctx.handle, 1, k._lib.ptr(o2), C.byref(inf), None) == 0
            assert pyref.point_from_wire(out) == want, (n, "with another MSM in flight")
            assert pyref.point_from_wire(o2) == want_other
    finally:
        srs.close()
