"""-m gpu: run-twice bit-compare (the reference's results are canonical field / group elements, so every entry point must
return the same bytes on every run): the same inputs through a FRESH context and SRS a second time, through the synchronous and the
asynchronous (slot) forms, and with other work in flight beside them -- atomics-ordered sorts, equal-split partial sums and shuffle
trees may reorder additions, but never change a result."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import pyref
from pyref import R_

pytestmark = pytest.mark.gpu
TAU = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % R_


def _run_all(k, seed):
    ctx = k.Context(0)
    srs = k.SRS.generate(TAU, 1 << 18, ctx=ctx)
    lib = k._lib.load()
    rng = np.random.default_rng(seed)
    out = {}

    def scal(n):
        a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return a

    kz = k.KZG.new(ctx)
    for log_n in (11, 14, 18):
        n = 1 << log_n
        sc = scal(n)
        o = np.zeros(8, np.uint64); inf = C.c_uint8(0)
        assert lib.kzg_msm_g1_srs(ctx.handle, srs.handle, 0, k._lib.ptr(sc), n, k._lib.ptr(o), C.byref(inf)) == 0
        out["msm%d" % log_n] = o.copy()
        # the same MSM through two slots at once (the second copy runs beside the first)
        assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, 0, k._lib.ptr(sc), n, 1) == 0
        assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, 0, k._lib.ptr(sc), n, 2) == 0
        o1 = np.zeros(8, np.uint64); o2 = np.zeros(8, np.uint64)
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 1, k._lib.ptr(o1), C.byref(inf), None) == 0
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 2, k._lib.ptr(o2), C.byref(inf), None) == 0
        assert np.array_equal(o1, o) and np.array_equal(o2, o), log_n
        poly = k.PolynomialEvalForm(sc)
        kz.calculate_and_store_roots_of_unity(n * 32)
        proof, y = kz._compute_proof_impl(poly, sc[7], srs, want_y=True)
        out["proof%d" % log_n] = np.concatenate([proof, y])
        f = sc.copy()
        assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(f), n, 0) == 0
        out["ntt%d" % log_n] = f
    out["ifft64"] = np.asarray(kz.g1_ifft(64, srs))
    raw = rng.integers(32, 127, size=40000, dtype=np.uint8).tobytes()
    blob = k.Blob.from_raw_data(raw)
    kz.calculate_and_store_roots_of_unity(len(blob))
    c, p, z, y = kz.commit_and_prove_blob(blob, srs)
    out["blob"] = np.concatenate([c, p, z, y])
    zs, ys = k.helpers.compute_challenges_and_evaluate_polynomial([blob] * 3, [c] * 3, ctx)
    out["batch_eval"] = np.concatenate([np.stack(zs).ravel(), np.stack(ys).ravel()])
    srs.close(); ctx.close()
    return out


def test_every_entry_point_returns_the_same_bytes_on_a_second_run():
    import rust_kzg_bn254_amd as k
    k.load()
    a = _run_all(k, 20261004)
    b = _run_all(k, 20261004)
    assert a.keys() == b.keys()
    for key in a:
        assert np.array_equal(a[key], b[key]), key
