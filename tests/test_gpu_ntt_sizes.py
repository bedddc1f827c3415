"""-m gpu: every pass shape of the Fr NTT (csrc/ntt.hip) against the oracle, and the proof / commitment path on the sizes whose
transforms run K = 9 tiles.

P = ceil(log n / 10) passes of K = ceil(log n / P) stages each: log n = 9, 17, 18, 19 run K = 9 tiles (4 columns, the `spread = 8`
branch of the XOR swizzle), 21 runs 7 + 7 + 7, >= 2^23 takes the lookup-twiddle fallback instead of the per-element twiddle array.
2^18 / 2^19 are the reference's large-blob sizes (prover/benches/bench_kzg_commit_large_blobs.rs:17-37); the reference accepts
domains up to 2^28 (primitives/src/polynomial.rs:42, :130-140, :241-251).
"""
import hashlib
import os
import random
import subprocess
import sys

import numpy as np
import pytest

import oracle as orc
import pyref
from pyref import R_

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TAU = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % R_
MONT = (1 << 256) % R_


@pytest.fixture(scope="module")
def k():
    import rust_kzg_bn254_amd as k
    k.load()
    k.default_context()
    return k


def ints_to_mont(vals):
    buf = b"".join((v * MONT % R_).to_bytes(32, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def random_canonical(n, seed):
    """n canonical Montgomery residues: uniform 252-bit values (< r)."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return a


@pytest.mark.parametrize("log_n", list(range(0, 23)))
def test_ntt_every_size_matches_oracle(k, log_n):
    """Both directions at EVERY log n from 0 to 22, bit for bit against the oracle's radix-2 transform."""
    n = 1 << log_n
    a = random_canonical(n, 7000 + log_n)
    ctx = k.default_context(); lib = k._lib.load()
    for inverse in (0, 1):
        got = a.copy()
        assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(got), n, inverse) == 0
        want = orc.fr_ntt_mt(a, inverse=bool(inverse)) if log_n >= 14 else orc.fr_ntt(a, inverse=bool(inverse))
        assert np.array_equal(got, want), (log_n, inverse)


@pytest.mark.parametrize("log_n", [23, 24])
def test_ntt_lookup_twiddle_fallback_matches_oracle(k, log_n):
    """Transforms above 2^22 read their inter-pass twiddles through the two-level lookup: oracle bit-compare in
    both directions, exact round trip, and the definition F[i] = sum_j a_j w^(ij) at sparse indices."""
    n = 1 << log_n
    a = random_canonical(n, 9000 + log_n)
    ctx = k.default_context(); lib = k._lib.load()
    f = a.copy()
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(f), n, 0) == 0
    assert np.array_equal(f, orc.fr_ntt_mt(a, inverse=False)), log_n
    back = f.copy()
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(back), n, 1) == 0
    assert np.array_equal(back, a)
    inv = a.copy()
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(inv), n, 1) == 0
    assert np.array_equal(inv, orc.fr_ntt_mt(a, inverse=True)), log_n
    del f, back, inv
    sparse = np.zeros((n, 4), np.uint64)
    idxs = [0, 1, 5_000_001, n - 1]
    vals = [3, 5, 7, 11]
    for j, v in zip(idxs, vals):
        sparse[j] = pyref.fr_to_mont(v)
    assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(sparse), n, 0) == 0
    w = pyref.root_of_unity(log_n)
    for i in (0, 1, 2, 1234567, n // 2, n // 2 + 1, n - 1):
        want = sum(v * pow(w, i * j, R_) for j, v in zip(idxs, vals)) % R_
        assert pyref.fr_from_mont(sparse[i]) == want, (log_n, i)


class Domain:
    def __init__(self, log_n):
        self.n = 1 << log_n
        w = pyref.root_of_unity(log_n) if log_n else 1
        self.roots, cur = [], 1
        for _ in range(self.n):
            self.roots.append(cur)
            cur = cur * w % R_

    def evaluate(self, evals, x):
        """(x^n - 1)/n * sum_i f_i w^i / (x - w^i), x off the domain (primitives/src/helpers.rs:507-532)."""
        dens = [(x - w) % R_ for w in self.roots]
        pre, acc = [], 1
        for d in dens:
            pre.append(acc)
            acc = acc * d % R_
        inv = pow(acc, -1, R_)
        tot = 0
        for i in range(self.n - 1, -1, -1):
            tot += evals[i] * self.roots[i] % R_ * (inv * pre[i] % R_)
            inv = inv * dens[i] % R_
        return tot % R_ * (pow(x, self.n, R_) - 1) % R_ * pow(self.n, -1, R_) % R_


@pytest.fixture(scope="module")
def srs19(k):
    s = k.SRS.generate(TAU, 1 << 19)
    yield s
    s.close()


@pytest.mark.parametrize("log_n", [9, 17, 18, 19])
def test_commit_and_proofs_on_k9_transform_sizes(k, srs19, log_n):
    """commit_eval_form (kzg.rs:84-104) and compute_proof_impl (kzg.rs:128-178, :237-260), off the domain and on it, at the sizes
    whose INTT runs K = 9 tiles; 2^18 / 2^19 are the reference's 8 MB / 16 MB blobs.  Expected values by big integers on the
    known-tau SRS: commitment = f^(tau) G1, proof = ((f^(tau) - y) / (tau - z)) G1."""
    n = 1 << log_n
    rnd = random.Random(0xBEEF + log_n)
    evals = [rnd.randrange(R_) for _ in range(n)]
    dom = Domain(log_n)
    ftau = dom.evaluate(evals, TAU)
    poly = k.PolynomialEvalForm(ints_to_mont(evals))
    kzg = k.KZG.new(); kzg.calculate_and_store_roots_of_unity(n * 32)
    commitment = kzg.commit_eval_form(poly, srs19)
    assert pyref.point_from_wire(commitment) == pyref.ec_mul(ftau, (1, 2)), log_n
    # coefficient form through the library's INTT == the oracle's, and commits to the same point
    coeffs = poly.to_coeff_form()
    assert np.array_equal(coeffs.coeffs(), orc.fr_ntt_mt(poly.evaluations(), inverse=True)), log_n
    assert np.array_equal(kzg.commit_coeff_form(coeffs, srs19), commitment)
    z_off = rnd.randrange(R_)
    cases = [(z_off, dom.evaluate(evals, z_off))] + [(dom.roots[m], evals[m]) for m in sorted({0, n // 2 + 1, n - 1})]
    for z, y_want in cases:
        proof, y = kzg._compute_proof_impl(poly, pyref.fr_to_mont(z), srs19, want_y=True)
        assert pyref.fr_from_mont(y) == y_want, (log_n, z == z_off)
        want_pt = pyref.ec_mul((ftau - y_want) * pow(TAU - z, -1, R_) % R_, (1, 2))
        assert pyref.point_from_wire(proof) == want_pt, (log_n, z == z_off)


def test_randomised_soak_of_commit_and_proof():
    """tools/soak_proof.py for four seconds with a fixed seed: random domain sizes 2^0 .. 2^14 (2^9 among them), dense / sparse /
    few-valued evaluations, z on and off the domain, every result against big-integer arithmetic."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SOAK_SECONDS="4", SOAK_SEED="20261004")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_proof.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "soak ok" in r.stdout, r.stdout[-400:] + r.stderr[-400:]


NTT_TILE_CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import oracle as orc
import rust_kzg_bn254_amd as k
k.load(); ctx = k.default_context(); lib = k._lib.load()
for log_n in range(0, 23):
    n = 1 << log_n
    rng = np.random.default_rng(4000 + log_n)
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
    for inverse in (0, 1):
        got = a.copy()
        assert lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(got), n, inverse) == 0
        want = orc.fr_ntt_mt(a, inverse=bool(inverse)) if log_n >= 14 else orc.fr_ntt(a, inverse=bool(inverse))
        assert np.array_equal(got, want), (log_n, inverse)
print("tile %%s ok" %% os.environ["KZG_NTT_TILE_LOG"])
'''


@pytest.mark.parametrize("tile_log", [10, 11])
def test_both_ntt_tile_sizes_at_every_transform_size(tile_log):
    """The pass kernel exists for tiles of 1 024 and of 2 048 elements and the library picks one by transform size (ntt.hip
    ntt_small_tile_pays); here each is FORCED for every log n 0 .. 22 (KZG_NTT_TILE_LOG, read when the library loads: a child process
    per tile size) and compared with the oracle in both directions -- sizes the default never runs on that tile included."""
    env = dict(os.environ, KZG_NTT_TILE_LOG=str(tile_log))
    res = subprocess.run([sys.executable, "-c", NTT_TILE_CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0 and ("tile %d ok" % tile_log) in res.stdout, (res.stdout[-500:], res.stderr[-2000:])
