"""-m gpu: blob -> commitment + proof as a stream of jobs (`kzg_commit_and_prove_blob_begin` / `_end`, csrc/blobstream.hip).

Reference: KZG::commit_blob (prover/src/kzg.rs:182-185) + KZG::compute_blob_proof (kzg.rs:288-309) with the Fiat-Shamir point of
helpers::compute_challenge (primitives/src/helpers.rs:411-472).  Expected values: the CPU oracle (transcript, proof) at 2^12 and below,
plain big-integer arithmetic on the known-tau SRS at 2^20, and -- for job schedules -- the one-call entry, itself checked the same way in
tests/test_gpu_config4.py.  Bit-exact everywhere.
"""
import ctypes as C
import hashlib
import random

import numpy as np
import pytest

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


def make_blob(k, rnd, length, canonical=False):
    raw = bytearray(rnd.randrange(256) for _ in range(length))
    if canonical:
        for i in range(0, length, 32):
            raw[i] &= 0x1F
    raw = bytes(raw)
    return raw, k.Blob.from_padded_unchecked(raw)


def kzg_for(k, length):
    kz = k.KZG.new()
    kz.calculate_and_store_roots_of_unity(length)
    return kz


def oracle_expect(orc, srs, raw, n, commitment):
    z = orc.compute_challenge(raw, commitment)
    evals = orc.to_fr_array(raw)
    evals = np.concatenate([evals, np.zeros((n - len(evals), 4), np.uint64)])
    cnt, roots = orc.calculate_roots_of_unity(len(raw))
    assert cnt == n
    rc, proof, y = orc.compute_proof(srs.g1, evals, roots, z, literal=False)
    assert rc == 0
    return z, y, proof


@pytest.mark.parametrize("cached", [False, True])
def test_stream_2_12_against_oracle(k, cached):
    """Six blobs of 2^12 elements (ragged tails, chunks >= r) through four jobs: every commitment, challenge, y and proof equals the oracle's."""
    import oracle as orc
    n = 1 << 12
    srs = k.SRS.generate(TAU, n)
    if cached:
        srs.cache_lagrange(n)
    rnd = random.Random(612 + cached)
    blobs = [make_blob(k, rnd, 32 * n - t) for t in (0, 9, 31, 32 * 100 + 5, 1, 17)]
    kz = kzg_for(k, 32 * n)
    got = list(kz.commit_and_prove_blobs([b for _, b in blobs], srs, inflight=4))
    assert len(got) == len(blobs)
    for (raw, blob), (com, proof, z, y) in zip(blobs, got):
        rc, want_c = orc.commit_eval_form(srs.g1, np.concatenate([orc.to_fr_array(raw), np.zeros((n - (len(raw) + 31) // 32, 4), np.uint64)]), literal=False)
        assert rc == 0 and np.array_equal(com, want_c)
        wz, wy, wp = oracle_expect(orc, srs, raw, n, com)
        assert np.array_equal(z, wz) and np.array_equal(y, wy) and np.array_equal(proof, wp)
    srs.close()


def test_stream_mixed_sizes_and_given_commitments(k):
    """Jobs of different lengths (1 element ... 2^14) interleaved, some with the caller's commitment (KZG::compute_blob_proof as it stands):
    identical to the one-call entries, whatever the order of the end calls."""
    srs = k.SRS.generate(TAU, 1 << 14)
    srs.cache_lagrange(1 << 13)                       # one size over the cached basis, the others through IFFT + MSM
    rnd = random.Random(77)
    lens = [5, 32, 33, 32 * 64, 32 * 200 + 7, 32 * 1024, 32 * 4096 - 3, 32 * 8192, 32 * 8192 - 31, 32 * 16384, 32 * 3000, 32 * 16384 - 1]
    cases = []
    for i, ln in enumerate(lens):
        raw, blob = make_blob(k, rnd, ln)
        kz = kzg_for(k, ln)
        com, proof, z, y = kz.commit_and_prove_blob(blob, srs)
        cases.append((kz, blob, com, proof, z, y, i % 3 == 1))
    order = list(range(len(cases)))
    for rep in range(3):
        for i in order:
            kz, blob, com, _p, _z, _y, given = cases[i]
            kz.commit_and_prove_blob_begin(blob, srs, i, commitment=com if given else None)
        rnd.shuffle(order)
        for i in order:
            kz, blob, com, proof, z, y, given = cases[i]
            c2, p2, z2, y2 = kz.commit_and_prove_blob_end(i)
            assert np.array_equal(c2, com) and np.array_equal(p2, proof) and np.array_equal(z2, z) and np.array_equal(y2, y), (rep, i)
    srs.close()


def test_stream_guards(k):
    lib = k._lib.load()
    n = 1 << 10
    srs = k.SRS.generate(TAU, n)
    rnd = random.Random(5)
    raw, blob = make_blob(k, rnd, 32 * n)
    kz = kzg_for(k, 32 * n)
    with pytest.raises(k.errors.GenericError, match="inconsistent length between blob and root of unities"):
        kzg_for(k, 32 * 64).commit_and_prove_blob_begin(blob, srs, 0)
    with pytest.raises(k.errors.NotOnCurveError):
        kz.commit_and_prove_blob_begin(blob, srs, 0, commitment=pyref.point_to_wire((1, 3)))
    short = k.SRS.generate(TAU, n // 2)
    with pytest.raises(k.errors.SrsCapacityExceeded):
        kz.commit_and_prove_blob_begin(blob, short, 0)
    short.close()
    with pytest.raises(ValueError):
        kz.commit_and_prove_blob_begin(blob, srs, k._lib.BLOB_JOBS)
    with pytest.raises(ValueError):
        kz.commit_and_prove_blob_end(3)                                   # nothing in flight there
    kz.commit_and_prove_blob_begin(blob, srs, 2)
    with pytest.raises(ValueError):
        kz.commit_and_prove_blob_begin(blob, srs, 2)                      # in flight
    want = kz.commit_and_prove_blob_end(2)
    # a slot held by one of the caller's own asynchronous calls is left alone: the jobs work around it
    scal = np.frombuffer(bytes(rnd.randrange(256) for _ in range(32 * n)), dtype=np.uint64).copy().reshape(n, 4)
    scal[:, 3] &= (1 << 60) - 1
    assert lib.kzg_msm_g1_srs_begin(kz._ctx().handle, srs.handle, 0, k._lib.ptr(scal), n, 1) == 0
    for j in range(6):
        kz.commit_and_prove_blob_begin(blob, srs, j)
    for j in range(6):
        got = kz.commit_and_prove_blob_end(j)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
    out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
    assert lib.kzg_msm_g1_srs_end(kz._ctx().handle, 1, k._lib.ptr(out), C.byref(inf), None) == 0
    rc = lib.kzg_msm_g1_srs(kz._ctx().handle, srs.handle, 0, k._lib.ptr(scal), n, k._lib.ptr(np.zeros(8, np.uint64)), C.byref(inf), None)
    assert rc == 0
    # jobs left in flight are joined by the context's destruction
    ctx2 = k._lib.Context(0)
    srs2 = k.SRS.generate(TAU, n, ctx=ctx2)
    buf = np.frombuffer(raw, dtype=np.uint8)
    assert lib.kzg_commit_and_prove_blob_begin(ctx2.handle, srs2.handle, buf.ctypes.data_as(k._lib.u8p), len(raw), n, None, 0) == 0
    srs2.close()
    ctx2.close()
    srs.close()


def test_stream_2_20_known_tau(k):
    """32 MiB blobs, eight jobs in flight over the cached Lagrange basis: commitment == f^(tau) G1, z == the oracle's transcript, y and the proof
    by big-integer arithmetic for one blob, the one-call entry for the other; the repeats must reproduce them bit for bit."""
    import oracle as orc
    from test_gpu_config4 import Domain, expect_point, proof_scalar
    log_n = 20
    n = 1 << log_n
    dom = Domain(log_n)
    srs = k.SRS.generate(TAU, n)
    srs.cache_lagrange(n)
    rng = np.random.default_rng(2026)
    datas = []
    for _ in range(2):
        raw = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        raw[:, 0] &= 0x1F
        datas.append(raw.tobytes())
    blobs = [k.Blob.from_padded_unchecked(d) for d in datas]
    kz = kzg_for(k, 32 * n)
    seq = [0, 1, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1]
    got = list(kz.commit_and_prove_blobs((blobs[i] for i in seq), srs, inflight=8))
    first = {}
    for i, res in zip(seq, got):
        if i in first:
            assert all(np.array_equal(a, b) for a, b in zip(res, first[i]))
            continue
        first[i] = res
        com, proof, z, y = res
        if i == 1:                                      # the second blob: against the one-call entry (tests/test_gpu_config4.py checks that one by big integers)
            assert all(np.array_equal(a, b) for a, b in zip(res, kz.commit_and_prove_blob(blobs[1], srs)))
            continue
        data = datas[i]
        evals = [int.from_bytes(data[32 * t:32 * t + 32], "big") for t in range(n)]
        ftau = dom.evaluate(evals, TAU)
        assert pyref.point_from_wire(com) == expect_point(ftau)
        assert np.array_equal(z, orc.compute_challenge(data, com))
        zi = pyref.fr_from_mont(z)
        yi = dom.evaluate(evals, zi)
        assert pyref.fr_from_mont(y) == yi
        assert pyref.point_from_wire(proof) == expect_point(proof_scalar(ftau, yi, zi))
    srs.close()


def test_randomised_soak_of_the_blob_stream():
    """tools/soak_blob_stream.py for four seconds with a fixed seed (profiles/r06_soak.txt holds the long runs): random lengths, 1 .. 16 jobs in flight, shuffled end
    order, given commitments, cached bases, a slot held by the caller -- every job bit for bit against the one-call entries."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SOAK_SECONDS="4", SOAK_SEED="20261005", SOAK_MAX_LOG="14")
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_blob_stream.py")], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0 and "blob stream soak ok" in res.stdout, (res.stdout[-500:], res.stderr[-2000:])
