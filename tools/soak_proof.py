"""Randomised soak of compute_proof / commit_eval_form against big-integer arithmetic on the known-tau SRS: random domain sizes 2^0..2^14,
random evaluations (dense, sparse, few values), z off the domain and on it.  SOAK_SECONDS (default 40)."""
import hashlib, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import pyref
from pyref import R_
import rust_kzg_bn254_amd as k
k.load(); k.default_context()
TAU = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % R_
srs = k.SRS.generate(TAU, 1 << 14)
seed = int(os.environ.get("SOAK_SEED", str(int(time.time())))); rnd = random.Random(seed)
print("seed", seed, flush=True)
doms = {}
def domain(log_n):
    if log_n not in doms:
        w = pyref.root_of_unity(log_n) if log_n else 1
        r, cur = [], 1
        for _ in range(1 << log_n): r.append(cur); cur = cur * w % R_
        doms[log_n] = r
    return doms[log_n]
def bary(evals, roots, x):
    n = len(evals)
    if n == 1: return evals[0]
    dens = [(x - r) % R_ for r in roots]
    pre, acc = [], 1
    for d in dens: pre.append(acc); acc = acc * d % R_
    inv = pow(acc, -1, R_); tot = 0
    for i in range(n - 1, -1, -1):
        tot += evals[i] * roots[i] % R_ * (inv * pre[i] % R_); inv = inv * dens[i] % R_
    return tot % R_ * (pow(x, n, R_) - 1) % R_ * pow(n, -1, R_) % R_
t_end = time.time() + float(os.environ.get("SOAK_SECONDS", "40")); cases = 0
while time.time() < t_end:
    log_n = rnd.choice([0, 1, 2, 3, 4, 6, 8, 9, 10, 11, 12, 12, 13, 13, 14]); n = 1 << log_n
    roots = domain(log_n)
    kind = rnd.randrange(3)
    if kind == 0: evals = [rnd.randrange(R_) for _ in range(n)]
    elif kind == 1: evals = [rnd.randrange(R_) if rnd.random() < 0.05 else 0 for _ in range(n)]
    else:
        few = [rnd.randrange(R_), 1, R_ - 1, 0]; evals = [rnd.choice(few) for _ in range(n)]
    poly = k.PolynomialEvalForm(pyref.frs_to_mont(evals))
    kz = k.KZG.new(); kz.calculate_and_store_roots_of_unity(n * 32)
    ftau = bary(evals, roots, TAU)
    c = kz.commit_eval_form(poly, srs)
    want_c = pyref.ec_mul(ftau, (1, 2)) if ftau else None
    assert pyref.point_from_wire(c) == want_c, ("commit", seed, cases, log_n)
    on = rnd.random() < 0.5
    m = rnd.randrange(n)
    z = roots[m] if on else rnd.randrange(R_)
    y_want = evals[m] if on else bary(evals, roots, z)
    proof, y = kz._compute_proof_impl(poly, pyref.fr_to_mont(z), srs, want_y=True)
    assert pyref.fr_from_mont(y) == y_want, ("y", seed, cases, log_n, on)
    qs = (ftau - y_want) * pow(TAU - z, -1, R_) % R_
    want_p = pyref.ec_mul(qs, (1, 2)) if qs and n > 1 else None
    assert pyref.point_from_wire(proof) == want_p, ("proof", seed, cases, log_n, on)
    cases += 1
print("soak ok:", cases, "cases", flush=True)
