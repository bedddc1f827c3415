"""-m gpu: BASELINE config 4 ("Full blob -> commit + proof, 2^20 SRS, sharded MSM") at FULL size on the known-tau SRS.

Every expected value here comes from plain Python big-integer arithmetic on the inputs (barycentric evaluation of the
evaluation-form polynomial at tau and at z, one scalar multiplication of G1) -- never from another run of the HIP path:
    commit_eval_form(f)      == f^(tau) * G1                          prover/src/kzg.rs:84-104
    compute_proof(f, z)      == ((f^(tau) - y) / (tau - z)) * G1      prover/src/kzg.rs:128-178 (off-domain z)
    compute_proof(f, w^m)    == the same with y = f_m                 prover/src/kzg.rs:237-260 (on-domain branch)
    commit_blob(32 MiB)      == the same from raw bytes               prover/src/kzg.rs:182-185, primitives/src/helpers.rs:40-57
and the same four through the multi-GPU entry points (kzg_commit_eval_form_partial / kzg_compute_proof_partial) with 8
uneven SRS shards folded by kzg_g1_fold_partials (verifier/tests/tests.rs:79-132 is the reference's end-to-end shape).
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
LOG_N = 20
N = 1 << LOG_N
G = (1, 2)
MONT = (1 << 256) % R_


def ints_to_mont(vals):
    """python ints -> (n, 4) uint64 wire array (Montgomery, R = 2^256)."""
    buf = b"".join((v * MONT % R_).to_bytes(32, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def batch_inverse(xs):
    pre, acc = [], 1
    for x in xs:
        pre.append(acc)
        acc = acc * x % R_
    inv = pow(acc, -1, R_)
    out = [0] * len(xs)
    for i in range(len(xs) - 1, -1, -1):
        out[i] = inv * pre[i] % R_
        inv = inv * xs[i] % R_
    return out


class Domain:
    """[1, w, .., w^(n-1)] with w = 5^((r-1)/n) and the barycentric evaluation f^(x) of an evaluation-form polynomial."""

    def __init__(self, log_n):
        self.n = 1 << log_n
        w = pyref.root_of_unity(log_n)
        self.roots, cur = [], 1
        for _ in range(self.n):
            self.roots.append(cur)
            cur = cur * w % R_

    def evaluate(self, evals, x):
        """(x^n - 1)/n * sum_i f_i w^i / (x - w^i), x not in the domain (helpers.rs:507-532)."""
        inv = batch_inverse([(x - w) % R_ for w in self.roots])
        s = 0
        for f, w, iv in zip(evals, self.roots, inv):
            s += f * w % R_ * iv
        return s % R_ * (pow(x, self.n, R_) - 1) % R_ * pow(self.n, -1, R_) % R_


@pytest.fixture(scope="module")
def k():
    import rust_kzg_bn254_amd as k
    k.load()
    k.default_context()
    return k


@pytest.fixture(scope="module")
def dom():
    return Domain(LOG_N)


@pytest.fixture(scope="module")
def case(dom):
    """One random evaluation-form polynomial of 2^20 elements with its big-integer ground truth."""
    rnd = random.Random(0xC0F4)
    evals = [rnd.randrange(R_) for _ in range(N)]
    ftau = dom.evaluate(evals, TAU)
    z = rnd.randrange(R_)
    y = dom.evaluate(evals, z)
    m = 777_777
    return {"evals": evals, "wire": ints_to_mont(evals), "ftau": ftau, "z": z, "y": y, "m": m}


@pytest.fixture(scope="module")
def srs(k):
    s = k.SRS.generate(TAU, N)
    yield s
    s.close()


def expect_point(scalar):
    return pyref.ec_mul(scalar % R_, G)


def proof_scalar(ftau, y, z):
    return (ftau - y) * pow(TAU - z, -1, R_) % R_


def test_commit_eval_form_2_20(k, srs, case):
    kz = k.KZG.new()
    c = kz.commit_eval_form(k.PolynomialEvalForm(case["wire"]), srs)
    assert pyref.point_from_wire(c) == expect_point(case["ftau"])


def test_compute_proof_2_20_off_domain(k, srs, case):
    kz = k.KZG.new()
    kz.calculate_and_store_roots_of_unity(N * 32)
    proof, y = kz._compute_proof_impl(k.PolynomialEvalForm(case["wire"]), pyref.fr_to_mont(case["z"]), srs, want_y=True)
    assert pyref.fr_from_mont(y) == case["y"]
    assert pyref.point_from_wire(proof) == expect_point(proof_scalar(case["ftau"], case["y"], case["z"]))


def test_compute_proof_2_20_on_domain(k, srs, case, dom):
    kz = k.KZG.new()
    kz.calculate_and_store_roots_of_unity(N * 32)
    m = case["m"]
    assert pyref.fr_from_mont(kz.get_nth_root_of_unity(m)) == dom.roots[m]
    proof = kz.compute_proof_with_known_z_fr_index(k.PolynomialEvalForm(case["wire"]), m, srs)
    assert pyref.point_from_wire(proof) == expect_point(proof_scalar(case["ftau"], case["evals"][m], dom.roots[m]))


def test_commit_blob_32mib(k, srs, dom):
    """32 MiB of bytes (2^20 chunks of 32, some of them >= r so that the mod-r reduction of helpers.rs:40-57 is exercised)."""
    rng = np.random.default_rng(404)
    raw = rng.integers(0, 256, size=(N, 32), dtype=np.uint8)
    raw[::3, 0] = 0                                   # a third of the chunks canonical (< 2^248), the rest arbitrary
    data = raw.tobytes()
    evals = [int.from_bytes(data[32 * i:32 * i + 32], "big") % R_ for i in range(N)]
    ftau = dom.evaluate(evals, TAU)
    kz = k.KZG.new()
    blob = k.Blob.from_padded_unchecked(data)
    assert pyref.point_from_wire(kz.commit_blob(blob, srs)) == expect_point(ftau)
    got = list(kz.commit_blob_stream([blob, blob], srs))
    assert all(pyref.point_from_wire(g) == expect_point(ftau) for g in got)


def test_sharded_commit_and_proof_2_20_eight_uneven_shards(k, case, dom):
    """kzg_commit_eval_form_partial / kzg_compute_proof_partial: 8 uneven shards of the SRS (one of them a single point),
    each generated with first_power, partials folded on the host -> big-integer expectation."""
    from rust_kzg_bn254_amd.sharding import fold_partials
    lib = k._lib.load()
    ctx = k.default_context()
    bounds = [0, 100_000, 100_001, 300_000, 524_288, 524_289 + 70_000, 800_000, 1_000_000, N]
    ev = np.ascontiguousarray(case["wire"])
    z_off = np.ascontiguousarray(pyref.fr_to_mont(case["z"]))
    m = case["m"]
    z_on = np.ascontiguousarray(pyref.fr_to_mont(dom.roots[m]))
    pcs, pps, pos = [], [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        shard = k.SRS.generate(TAU, hi - lo, first_power=lo, ctx=ctx)
        pc = np.zeros(16, np.uint64); pp = np.zeros(16, np.uint64); po = np.zeros(16, np.uint64); y = np.zeros(4, np.uint64)
        assert lib.kzg_commit_eval_form_partial(ctx.handle, shard.handle, lo, k._lib.ptr(ev), N, k._lib.ptr(pc)) == 0
        assert lib.kzg_compute_proof_partial(ctx.handle, shard.handle, lo, k._lib.ptr(ev), N, None, N, k._lib.ptr(z_off),
                                             k._lib.ptr(pp), k._lib.ptr(y)) == 0
        assert pyref.fr_from_mont(y) == case["y"]
        assert lib.kzg_compute_proof_partial(ctx.handle, shard.handle, lo, k._lib.ptr(ev), N, None, N, k._lib.ptr(z_on),
                                             k._lib.ptr(po), k._lib.ptr(y)) == 0
        assert pyref.fr_from_mont(y) == case["evals"][m]
        pcs.append(pc); pps.append(pp); pos.append(po)
        shard.close()
    assert pyref.point_from_wire(fold_partials(np.stack(pcs))) == expect_point(case["ftau"])
    assert pyref.point_from_wire(fold_partials(np.stack(pps))) == expect_point(proof_scalar(case["ftau"], case["y"], case["z"]))
    assert pyref.point_from_wire(fold_partials(np.stack(pos))) == expect_point(proof_scalar(case["ftau"], case["evals"][m], dom.roots[m]))


def test_sharded_msm_stream_2_20_folds_to_known_tau(k, case):
    """The bench's multi-rank shape on one GPU: 8 equal shards of a 2^20-pair MSM, each through the asynchronous slots
    (ShardedMsm.begin / kzg_msm_g1_srs_end partial), folded -> sum_i c_i tau^i * G1."""
    import torch
    from rust_kzg_bn254_amd.sharding import ShardedMsm, fold_partials
    ctx = k.default_context()
    coeffs = case["evals"]                              # used as coefficients here
    ptau, cur = 0, 1
    for v in coeffs:
        ptau = (ptau + v * cur) % R_
        cur = cur * TAU % R_
    parts = []
    world = 8
    for rank in range(world):
        sh = ShardedMsm(ctx, N, rank, world, gather_device=None)
        shard = k.SRS.generate(TAU, sh.len, first_power=sh.lo, ctx=ctx)
        d = torch.from_numpy(np.ascontiguousarray(case["wire"][sh.lo:sh.hi]).view(np.int64)).cuda()
        torch.cuda.synchronize()
        sh.begin(shard, d.data_ptr(), rank % k._lib.NUM_SLOTS)
        parts.append(sh._end_partial(rank % k._lib.NUM_SLOTS))
        shard.close()
    assert pyref.point_from_wire(fold_partials(np.stack(parts))) == expect_point(ptau)


# ---------------------------------------------------------------------------------------------------------
# KZG::compute_blob_proof / commit + proof in one call (prover/src/kzg.rs:288-309) behind the C-ABI
# ---------------------------------------------------------------------------------------------------------
def test_blob_proof_2_12_against_oracle(k):
    """2^12 elements: challenge == oracle transcript, proof == oracle proof; the fused commit + proof call returns the same."""
    import oracle as orc
    n = 1 << 12
    srs = k.SRS.generate(TAU, n)
    rnd = random.Random(12)
    raw = bytes(rnd.randrange(256) for _ in range(32 * n - 9))        # ragged tail, non-canonical chunks
    blob = k.Blob.from_padded_unchecked(raw) if hasattr(k.Blob, "from_padded_unchecked") else k.Blob(raw)
    kz = k.KZG.new()
    kz.calculate_and_store_roots_of_unity(len(raw))
    commitment = kz.commit_blob(blob, srs)
    proof, z, y = kz.compute_blob_proof(blob, commitment, srs, want_zy=True)
    assert np.array_equal(z, orc.compute_challenge(raw, commitment))
    evals = orc.to_fr_array(raw)
    evals = np.concatenate([evals, np.zeros((n - len(evals), 4), np.uint64)])
    rc, roots = orc.calculate_roots_of_unity(len(raw))
    rc, want, want_y = orc.compute_proof(srs.g1, evals, roots, z, literal=False)
    assert rc == 0 and np.array_equal(proof, want) and np.array_equal(y, want_y)
    c2, p2, z2, y2 = kz.commit_and_prove_blob(blob, srs)
    assert np.array_equal(c2, commitment) and np.array_equal(p2, proof) and np.array_equal(z2, z) and np.array_equal(y2, y)
    # guards: wrong root count, commitment off the curve, SRS too short
    kz2 = k.KZG.new(); kz2.calculate_and_store_roots_of_unity(32 * 64)
    with pytest.raises(k.errors.GenericError, match="inconsistent length between blob and root of unities"):
        kz2.compute_blob_proof(blob, commitment, srs)
    with pytest.raises(k.errors.NotOnCurveError):
        kz.compute_blob_proof(blob, pyref.point_to_wire((1, 3)), srs)
    short = k.SRS.generate(TAU, n // 2)
    with pytest.raises(k.errors.SrsCapacityExceeded):
        kz.compute_blob_proof(blob, commitment, short)
    short.close(); srs.close()


def test_blob_proof_2_20_known_tau(k, srs, dom):
    """32 MiB blob: z == oracle transcript (SHA-256 over 32 MiB), proof == ((f^(tau) - y) / (tau - z)) * G1 by big-integer arithmetic."""
    import oracle as orc
    rng = np.random.default_rng(2020)
    raw = rng.integers(0, 256, size=(N, 32), dtype=np.uint8)
    raw[:, 0] &= 0x1F                                                  # canonical chunks (< 2^253 < r)
    data = raw.tobytes()
    evals = [int.from_bytes(data[32 * i:32 * i + 32], "big") for i in range(N)]
    blob = k.Blob.from_padded_unchecked(data)
    kz = k.KZG.new()
    kz.calculate_and_store_roots_of_unity(len(data))
    com, proof, z, y = kz.commit_and_prove_blob(blob, srs)
    ftau = dom.evaluate(evals, TAU)
    assert pyref.point_from_wire(com) == expect_point(ftau)
    assert np.array_equal(z, orc.compute_challenge(data, com))
    zi = pyref.fr_from_mont(z)
    yi = dom.evaluate(evals, zi)
    assert pyref.fr_from_mont(y) == yi
    assert pyref.point_from_wire(proof) == expect_point(proof_scalar(ftau, yi, zi))
    assert np.array_equal(kz.compute_blob_proof(blob, com, srs), proof)


def test_small_and_mid_commitments_on_the_2_20_srs(k, srs):
    """One loaded 2^20-point SRS carries two window-table sets (srs.hip: c = 17, and c = 15 for MSMs of <= 2^13 pairs): coefficient-form
    commitments on either side of that limit, and MSMs over an SRS slice that does not start at 0, against p(tau) * G1 by big integers
    (prover/src/kzg.rs:107-124: the commitment uses the first len(poly) SRS points)."""
    lib = k._lib.load()
    ctx = k.default_context()
    rnd = random.Random(0x5A11)
    kz = k.KZG.new()
    for n in (1, 100, 2048, 8192, 8193, 40_000):
        coeffs = [rnd.randrange(R_) for _ in range(n)]
        want = 0
        for cf in reversed(coeffs):
            want = (want * TAU + cf) % R_
        got = kz.commit_coeff_form(k.PolynomialCoeffForm(ints_to_mont(coeffs)), srs)
        assert pyref.point_from_wire(got) == expect_point(want), n
    for offset, n in ((12_345, 4096), (777_000, 8192), (1_000_000, 20_000)):
        sc = [rnd.randrange(R_) for _ in range(n)]
        want, tp = 0, pow(TAU, offset, R_)
        for s_ in sc:
            want = (want + s_ * tp) % R_
            tp = tp * TAU % R_
        out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
        wire = ints_to_mont(sc)
        assert lib.kzg_msm_g1_srs(ctx.handle, srs.handle, offset, k._lib.ptr(wire), n, k._lib.ptr(out), C.byref(inf)) == 0
        assert pyref.point_from_wire(out) == expect_point(want), (offset, n)
