"""-m gpu: the reference's own upper bound.  Polynomials of up to 2^28 elements (primitives/src/polynomial.rs:42, :170; the domain
construction fails above it, :132-134) and SRS of up to 2^28 points (primitives/src/consts.rs:66, prover/src/srs.rs:36-40) are what the
reference accepts; tests/test_gpu_large_sizes.py stops at 2^26 because its inputs are host arrays.  Here everything large lives in HBM
(288 GB): 8 GiB of Fr elements, a 16 GiB SRS, 8 GiB of scalars -- the host holds a few thousand values.

* Fr NTT at 2^27 and 2^28 through kzg_fr_ntt_device: a polynomial with 96 random coefficients at random degrees (degree n - 1 among
  them) -> its n evaluations must equal sum_j c_j w^(i d_j) at 300 sampled indices (big-integer arithmetic on the host: independent of
  the transform), and the inverse transform must return the sparse vector exactly, all n elements compared on the device.
* a commitment over EVERY point of a 2^27- and a 2^28-point SRS (table-less generic mode, 8 / 16 launches of 2^24 pairs) with
  full-width scalars of period 4 096, against the closed form (sum_m k_m tau^m)(tau^n - 1)/(tau^4096 - 1) G1.
* one element / one point more is refused with the reference's errors, before anything is allocated."""
import ctypes as C
import hashlib
import time

import numpy as np
import pytest

import pyref
from pyref import R_

pytestmark = pytest.mark.gpu
TAU = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % R_


@pytest.fixture(scope="module")
def k():
    import torch  # noqa: F401  (before the library: INTEGRATION.md section 5)
    import rust_kzg_bn254_amd as k
    k.load()
    k.default_context()
    return k


@pytest.mark.parametrize("log_n", [27, 28])
def test_ntt_at_2_27_and_2_28_on_resident_data(k, log_n):
    import torch
    ctx = k.default_context(); lib = k._lib.load()
    n = 1 << log_n
    rng = np.random.default_rng(2700 + log_n)
    degs = sorted(set([0, 1, n // 2, n - 1] + [int(v) for v in rng.integers(0, n, size=92)]))
    cs = [int.from_bytes(rng.bytes(40), "little") % R_ for _ in degs]
    d = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    d[torch.tensor(degs, device="cuda")] = torch.from_numpy(pyref.frs_to_mont(cs).view(np.int64)).cuda()
    sparse = d.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d.data_ptr()), n, 0) == 0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    w = pyref.root_of_unity(log_n)
    sample = sorted(set([0, 1, 2, n // 2, n - 2, n - 1] + [int(v) for v in rng.integers(0, n, size=294)]))
    got = d[torch.tensor(sample, device="cuda")].cpu().numpy().view(np.uint64)
    for row, i in zip(got, sample):
        wi = pow(w, i, R_)
        assert pyref.fr_from_mont(row) == sum(c * pow(wi, dj, R_) for c, dj in zip(cs, degs)) % R_, (log_n, i)
    t0 = time.perf_counter()
    assert lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d.data_ptr()), n, 1) == 0
    torch.cuda.synchronize()
    dti = time.perf_counter() - t0
    assert torch.equal(d, sparse), "inverse(forward(x)) != x at 2^%d" % log_n
    print("  Fr NTT 2^%d (%d GiB resident): forward %.1f ms (first call: tables included), inverse %.1f ms" % (log_n, n * 32 >> 30, dt * 1e3, dti * 1e3))
    t0 = time.perf_counter()
    assert lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d.data_ptr()), n, 0) == 0
    torch.cuda.synchronize()
    print("  Fr NTT 2^%d again: %.1f ms = %.0f GB/s algorithmic (64 B per element)" % (log_n, (time.perf_counter() - t0) * 1e3, n * 64 / (time.perf_counter() - t0) / 1e9))
    del d, sparse
    torch.cuda.empty_cache()


@pytest.mark.parametrize("log_srs", [27, 28])
def test_commitment_over_every_point_of_a_2_27_and_2_28_point_srs(k, log_srs):
    import torch
    ctx = k.default_context(); lib = k._lib.load()
    n = 1 << log_srs
    period = 4096
    rng = np.random.default_rng(300 + log_srs)
    ks = [int.from_bytes(rng.bytes(40), "little") % R_ for _ in range(period)]
    block = torch.from_numpy(pyref.frs_to_mont(ks).view(np.int64)).cuda()
    d_sc = block.repeat(n // period, 1).contiguous()
    assert d_sc.shape == (n, 4)
    acc, tp = 0, 1
    for v in ks:
        acc = (acc + v * tp) % R_
        tp = tp * TAU % R_
    geo = (pow(TAU, n, R_) - 1) * pow(pow(TAU, period, R_) - 1, -1, R_) % R_
    want = pyref.ec_mul(acc * geo % R_, (1, 2))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    srs = k.SRS.generate(TAU, n)
    t_srs = time.perf_counter() - t0
    try:
        out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
        t0 = time.perf_counter()
        rc = lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d_sc.data_ptr()), n, k._lib.ptr(out), C.byref(inf))
        dt = time.perf_counter() - t0
        assert rc == 0, (rc, lib.kzg_ctx_last_error(ctx.handle))
        print("  SRS 2^%d (%d GiB, set-up %.1f s): commitment over all %d points, scalars resident: %.0f ms (%.2f ms per 2^20 pairs)"
              % (log_srs, n * 64 >> 30, t_srs, n, dt * 1e3, dt * 1e3 / (n >> 20)))
        assert pyref.point_from_wire(out) == want, log_srs
        # a window in the middle: 2^26 points from 2^25 + 4096 on -- the offset form on the same handle
        off, m = (1 << 25) + period, 1 << 26
        rc = lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, off, C.c_void_p(d_sc.data_ptr()), m, k._lib.ptr(out), C.byref(inf))
        assert rc == 0
        geo_m = (pow(TAU, m, R_) - 1) * pow(pow(TAU, period, R_) - 1, -1, R_) % R_
        assert pyref.point_from_wire(out) == pyref.ec_mul(acc * geo_m % R_ * pow(TAU, off, R_) % R_, (1, 2))
        # one pair more than the SRS holds: the reference's length error, nothing computed
        assert lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 1, C.c_void_p(d_sc.data_ptr()), n, k._lib.ptr(out), C.byref(inf)) == k._lib.ERR_MSM_LENGTH_MISMATCH
    finally:
        srs.close()
        del d_sc
        torch.cuda.empty_cache()


def test_one_more_than_2_28_is_refused_before_any_allocation(k):
    ctx = k.default_context(); lib = k._lib.load()
    h = C.c_void_p()
    t = np.ascontiguousarray(pyref.fr_to_mont(TAU))
    assert lib.kzg_srs_generate(ctx.handle, k._lib.ptr(t), 0, (1 << 28) + 1, C.byref(h)) == k._lib.ERR_TOO_LARGE
    assert lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(8), 1 << 29, 0) == k._lib.ERR_DOMAIN          # (never dereferenced: refused on n)
    assert lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(8), (1 << 28) + 2, 0) == k._lib.ERR_NOT_POWER_OF_TWO
