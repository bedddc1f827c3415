"""Randomised soak of the evaluation-index shards of the Lagrange basis (csrc/lagrange.hip) against the one-GPU calls on the known-tau SRS: random
domain sizes 2^0 .. 2^13, random shard edges (empty shards, one-element shards, edges next to the evaluation point), evaluations dense / sparse / of
few values, z off the domain and on it, through the four-step C-ABI, the grouped and two-slot streams and the one-call forms.  Every result must be
bit-identical to kzg_commit_eval_form / kzg_compute_proof on the same inputs (themselves soaked against big integers by soak_proof.py).
SOAK_SECONDS (default 40), SOAK_SEED."""
import ctypes as C
import hashlib
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import pyref
from pyref import R_
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd.sharding import ShardedKzgLagrange, fold_partials

k.load()
ctx = k.default_context()
lib = k._lib.load()
P = k._lib.ptr
TAU = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % R_
srs = k.SRS.generate(TAU, 1 << 13)
seed = int(os.environ.get("SOAK_SEED", str(int(time.time()))))
rnd = random.Random(seed)
print("seed", seed, flush=True)
t_end = time.time() + float(os.environ.get("SOAK_SECONDS", "40"))
cases = 0
shard_cache = {}
while time.time() < t_end:
    log_n = rnd.choice([0, 1, 2, 3, 5, 7, 9, 10, 11, 12, 13])
    n = 1 << log_n
    kind = rnd.randrange(3)
    if kind == 0:
        evals = [rnd.randrange(R_) for _ in range(n)]
    elif kind == 1:
        evals = [rnd.randrange(R_) if rnd.random() < 0.05 else 0 for _ in range(n)]
    else:
        few = [rnd.randrange(R_), 1, R_ - 1, 0]
        evals = [rnd.choice(few) for _ in range(n)]
    wire = pyref.frs_to_mont(evals)
    kz = k.KZG.new(); kz.calculate_and_store_roots_of_unity(n * 32)
    on = rnd.random() < 0.5
    m = rnd.randrange(n)
    z = np.ascontiguousarray(kz.get_nth_root_of_unity(m)) if on else pyref.fr_to_mont(rnd.randrange(R_))
    want_c = kz.commit_eval_form(k.PolynomialEvalForm(wire), srs)
    want_p, want_y = kz._compute_proof_impl(k.PolynomialEvalForm(wire), z, srs, want_y=True)
    G = rnd.choice([1, 2, 3, 5, 8])
    cuts = sorted(rnd.choice([0, n, m, min(n, m + 1), rnd.randrange(n + 1)]) for _ in range(G - 1))
    edges = [0] + cuts + [n]
    bounds = list(zip(edges[:-1], edges[1:]))
    if len(shard_cache) > 64:                       # (between cases only: the handles of the case in hand stay alive)
        for s_ in shard_cache.values():
            s_.close()
        shard_cache.clear()
    shards = []
    for lo, hi in bounds:
        key = (log_n, lo, hi)
        if key not in shard_cache:
            shard_cache[key] = srs.lagrange_shard(n, lo, hi - lo)
        shards.append(shard_cache[key])
    cparts = np.zeros((G, 16), np.uint64); yparts = np.zeros((G, 8), np.uint64); parts = np.zeros((G, 32), np.uint64)
    y = np.zeros(4, np.uint64)
    mode = rnd.randrange(3)                         # 0: four steps, commitment separately; 1: commitment + proof on two slots; 2: grouped launch
    # pass 1: every rank's partial barycentric sum (one GPU: one rank after the other)
    for g, (sh, (lo, hi)) in enumerate(zip(shards, bounds)):
        ev = np.ascontiguousarray(wire[lo:hi])
        rc = lib.kzg_compute_proof_lagrange_begin(ctx.handle, sh.handle, lo, P(ev) if hi > lo else None, hi - lo, n, P(z), g % 4)
        assert rc == 0, ("begin", rc, ctx.last_error(), seed, cases, log_n, on, m, bounds)
        assert lib.kzg_compute_proof_lagrange_partial_y(ctx.handle, g % 4, P(yparts[g])) == 0
        assert lib.kzg_compute_proof_lagrange_abort(ctx.handle, g % 4) == 0
    assert lib.kzg_lagrange_fold_y(P(yparts), G, n, P(z), P(y)) == 0
    assert np.array_equal(y, want_y), ("y", seed, cases, log_n, on, bounds)
    for g, (sh, (lo, hi)) in enumerate(zip(shards, bounds)):
        ev = np.ascontiguousarray(wire[lo:hi])
        evp = P(ev) if hi > lo else None
        yp = np.zeros(8, np.uint64)
        grouped = mode == 2 and hi > lo and lib.kzg_srs_has_bit_tables(sh.handle, 1) and lib.kzg_msm_batch_capacity(hi - lo) >= 2
        if mode == 0:
            assert lib.kzg_commit_eval_form_lagrange_partial(ctx.handle, sh.handle, evp, hi - lo, P(cparts[g])) == 0
            assert lib.kzg_compute_proof_lagrange_begin(ctx.handle, sh.handle, lo, evp, hi - lo, n, P(z), 1) == 0
        elif grouped:
            assert lib.kzg_commit_and_prove_lagrange_begin(ctx.handle, sh.handle, lo, evp, hi - lo, n, P(z), 1, 1) == 0
        else:
            assert lib.kzg_commit_and_prove_lagrange_begin(ctx.handle, sh.handle, lo, evp, hi - lo, n, P(z), 3, 1) == 0
        assert lib.kzg_compute_proof_lagrange_partial_y(ctx.handle, 1, P(yp)) == 0 and np.array_equal(yp, yparts[g])
        assert lib.kzg_compute_proof_lagrange_continue(ctx.handle, 1, P(y)) == 0
        if mode == 0:
            assert lib.kzg_compute_proof_lagrange_end(ctx.handle, 1, P(parts[g])) == 0
        elif grouped:
            assert lib.kzg_commit_and_prove_lagrange_end(ctx.handle, 1, P(cparts[g]), P(parts[g])) == 0
        else:
            if hi > lo:
                assert lib.kzg_msm_g1_srs_end(ctx.handle, 3, None, None, P(cparts[g])) == 0
            assert lib.kzg_compute_proof_lagrange_end(ctx.handle, 1, P(parts[g])) == 0
    proof = np.zeros(8, np.uint64); inf = C.c_uint8(0)
    assert lib.kzg_lagrange_fold_proof(P(parts), G, n, P(z), P(proof), C.byref(inf)) == 0
    assert np.array_equal(fold_partials(cparts), want_c), ("commit", seed, cases, log_n, bounds, mode)
    assert np.array_equal(proof, want_p), ("proof", seed, cases, log_n, on, bounds, mode)
    if G == 1:                                      # the Python host's one-call and stream forms over the single shard
        sk = ShardedKzgLagrange(ctx, shards[0], n, gather_device=None)
        p2, y2 = sk.compute_proof(wire, z, want_y=True)
        assert np.array_equal(p2, want_p) and np.array_equal(y2, want_y), ("one call", seed, cases)
        for c3, p3, y3 in sk.commit_and_prove_stream([(wire, z)] * 3, grouped=rnd.choice([None, False])):
            assert np.array_equal(c3, want_c) and np.array_equal(p3, want_p) and np.array_equal(y3, want_y), ("stream", seed, cases)
    cases += 1
print("soak ok:", cases, "cases", flush=True)
