"""-m gpu: BASELINE config 4 sharded by EVALUATION index over the Lagrange basis (csrc/lagrange.hip; prover/src/kzg.rs:96-100, :128-178,
:237-260): every rank sees only its slice of the evaluations and of g1_ifft(srs).  Checked against
  * the reference's golden proofs (kzg.proof.eq.input, all 40 rows) with the shards cut from the reference's OWN Lagrange file
    (lagrangeG1SRS.txt) and from kzg_srs_lagrange_shard,
  * the CPU oracle (commitment, y, proof; on and off the domain) at 2^6 .. 2^11 with uneven, empty and one-point shards,
  * the one-GPU calls (kzg_compute_proof with and without a cached Lagrange basis),
  * big-integer closed forms at 2^20 through 8 uneven shards (tests/test_gpu_config4.py holds the same expectations for the old path).
"""
import ctypes as C
import hashlib
import os
import random

import numpy as np
import pytest

import oracle as orc
import pyref
from pyref import R_

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TAU = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % R_


@pytest.fixture(scope="module")
def k():
    import rust_kzg_bn254_amd as k
    k.load()
    k.default_context()
    return k


@pytest.fixture(scope="module")
def ref_srs(k, test_srs_wire):
    s = k.SRS(test_srs_wire)
    yield s
    s.close()


def sharded(k, lag_shards, bounds, n, wire, z_wire, slots=None):
    """commitment, proof and y through the four-step C-ABI over the given shards (all on one GPU, one after the other or, with
    `slots`, several proofs in flight); the exchanges are plain concatenations here."""
    lib = k._lib.load()
    ctx = k.default_context()
    P = k._lib.ptr
    z = np.ascontiguousarray(z_wire)
    G = len(lag_shards)
    cparts = np.zeros((G, 16), np.uint64)
    yparts = np.zeros((G, 8), np.uint64)
    parts = np.zeros((G, 32), np.uint64)
    for g, (sh, (lo, hi)) in enumerate(zip(lag_shards, bounds)):
        ev = np.ascontiguousarray(wire[lo:hi])
        assert lib.kzg_commit_eval_form_lagrange_partial(ctx.handle, sh.handle, P(ev) if hi > lo else None, hi - lo, P(cparts[g])) == 0
    for g, (sh, (lo, hi)) in enumerate(zip(lag_shards, bounds)):
        slot = (g % k._lib.NUM_SLOTS) if slots else 0
        ev = np.ascontiguousarray(wire[lo:hi])
        assert lib.kzg_compute_proof_lagrange_begin(ctx.handle, sh.handle, lo, P(ev) if hi > lo else None, hi - lo, n, P(z), slot) == 0
        assert lib.kzg_compute_proof_lagrange_partial_y(ctx.handle, slot, P(yparts[g])) == 0
        # the rank cannot go on before it knows y: park its state by finishing it later -- on one GPU that means one rank at a time,
        # so y comes from a first pass over all ranks (phase 1 only), then a second pass runs all four steps
        assert lib.kzg_compute_proof_lagrange_abort(ctx.handle, slot) == 0
    y = np.zeros(4, np.uint64)
    assert lib.kzg_lagrange_fold_y(P(yparts), G, n, P(z), P(y)) == 0
    for g, (sh, (lo, hi)) in enumerate(zip(lag_shards, bounds)):
        slot = (g % k._lib.NUM_SLOTS) if slots else 0
        ev = np.ascontiguousarray(wire[lo:hi])
        yp = np.zeros(8, np.uint64)
        assert lib.kzg_compute_proof_lagrange_begin(ctx.handle, sh.handle, lo, P(ev) if hi > lo else None, hi - lo, n, P(z), slot) == 0
        assert lib.kzg_compute_proof_lagrange_partial_y(ctx.handle, slot, P(yp)) == 0
        assert np.array_equal(yp, yparts[g])
        assert lib.kzg_compute_proof_lagrange_continue(ctx.handle, slot, P(y)) == 0
        assert lib.kzg_compute_proof_lagrange_end(ctx.handle, slot, P(parts[g])) == 0
    commitment = np.zeros(8, np.uint64); proof = np.zeros(8, np.uint64); inf = C.c_uint8(0)
    assert lib.kzg_g1_fold_partials(P(cparts), G, P(commitment), C.byref(inf)) == 0
    assert lib.kzg_lagrange_fold_proof(P(parts), G, n, P(z), P(proof), C.byref(inf)) == 0
    return commitment, proof, y


def cut(k, srs, n, bounds):
    return [srs.lagrange_shard(n, lo, hi - lo) for lo, hi in bounds]


def pairs(edges):
    return list(zip(edges[:-1], edges[1:]))


def test_golden_proofs_through_shards_of_the_references_lagrange_file(k, ref_srs, gettysburg):
    """All 40 rows of kzg.proof.eq.input (z = w^idx, kzg.rs:237-260) with the 64 evaluations split over five uneven shards -- once
    with the shards uploaded from the reference's own lagrangeG1SRS.txt, once cut from kzg_srs_lagrange_shard."""
    blob = k.Blob.from_raw_data(gettysburg)
    poly = blob.to_polynomial_eval_form()
    wire = np.ascontiguousarray(poly.evaluations())
    n = len(wire)
    assert n == 64
    lag_pts = [tuple(int(v) for v in ln.strip().split(",")) for ln in open(os.path.join(GOLDEN, "lagrangeG1SRS.txt")) if ln.strip()]
    lag_wire = pyref.points_to_wire(lag_pts)
    bounds = pairs([0, 1, 17, 32, 33, 64])
    from_file = [k.SRS(lag_wire[lo:hi]) for lo, hi in bounds]
    from_ifft = cut(k, ref_srs, n, bounds)
    for a, b in zip(from_file, from_ifft):
        assert np.array_equal(a.g1, b.g1)
    kzg = k.KZG.new(); kzg.calculate_and_store_roots_of_unity(len(blob))
    want_c = kzg.commit_eval_form(poly, ref_srs)
    rows = [ln.strip().split(",") for ln in open(os.path.join(GOLDEN, "kzg.proof.eq.input")) if ln.strip()]
    assert len(rows) == 40
    for j, (idx, x, y) in enumerate(rows):
        z = kzg.get_nth_root_of_unity(int(idx))
        c, proof, yy = sharded(k, from_file if j % 2 else from_ifft, bounds, n, wire, z, slots=(j % 3 == 0))
        assert pyref.point_from_wire(proof) == (int(x), int(y)), idx
        assert np.array_equal(c, want_c)
        assert np.array_equal(yy, wire[int(idx)])                       # helpers.rs:497-504
    for s in from_file + from_ifft:
        s.close()


@pytest.mark.parametrize("log_n,edges", [(6, [0, 64]), (7, [0, 5, 5, 128]), (9, [0, 100, 101, 300, 512]), (10, [0, 1024]),
                                         (11, [0, 1, 1023, 1025, 2000, 2048]), (11, [0, 2048])])
def test_sharded_commit_and_proof_against_the_oracle(k, ref_srs, test_srs_wire, log_n, edges):
    n = 1 << log_n
    rnd = random.Random(log_n * 1000 + len(edges))
    evals = [rnd.randrange(R_) for _ in range(n)]
    wire = pyref.frs_to_mont(evals)
    bounds = pairs(edges)
    shards = cut(k, ref_srs, n, bounds)
    rc, roots = orc.calculate_roots_of_unity(n * 32)
    rc, want_c = orc.commit_eval_form(test_srs_wire, wire, literal=False)
    kzg = k.KZG.new(); kzg.calculate_and_store_roots_of_unity(n * 32)
    zs = [pyref.fr_to_mont(rnd.randrange(R_)), pyref.fr_to_mont(3), np.ascontiguousarray(roots[0]), np.ascontiguousarray(roots[n - 1]),
          np.ascontiguousarray(roots[edges[1] % n]), np.ascontiguousarray(roots[(edges[1] - 1) % n]), np.ascontiguousarray(roots[rnd.randrange(n)])]
    for z in zs:
        rc, want_p, want_y = orc.compute_proof(test_srs_wire, wire, roots, z, literal=False)
        assert rc == 0
        c, proof, y = sharded(k, shards, bounds, n, wire, z, slots=True)
        assert np.array_equal(c, want_c)
        assert np.array_equal(y, want_y)
        assert np.array_equal(proof, want_p)
        one_gpu, y1 = kzg._compute_proof_impl(k.PolynomialEvalForm(wire), z, ref_srs, want_y=True)
        assert np.array_equal(one_gpu, want_p) and np.array_equal(y1, want_y)
    for s in shards:
        s.close()


def test_one_gpu_proofs_over_the_cached_lagrange_basis(k, test_srs_wire):
    """kzg_compute_proof / kzg_compute_proof_begin / kzg_compute_blob_proof / kzg_commit_blob with kzg_srs_cache_lagrange(n): the
    quotient's evaluations are committed over the Lagrange basis (kzg.rs:176-177 literally) instead of IFFT + monomial MSM -- the
    same points, checked against the oracle and against the uncached call on a second SRS handle."""
    srs_plain = k.SRS(test_srs_wire)
    srs_lag = k.SRS(test_srs_wire)
    rnd = random.Random(77)
    for log_n in (1, 5, 8, 11):
        n = 1 << log_n
        srs_lag.cache_lagrange(n)
        wire = pyref.frs_to_mont([rnd.randrange(R_) for _ in range(n)])
        rc, roots = orc.calculate_roots_of_unity(n * 32)
        kzg = k.KZG.new(); kzg.calculate_and_store_roots_of_unity(n * 32)
        poly = k.PolynomialEvalForm(wire)
        for z in (pyref.fr_to_mont(rnd.randrange(R_)), np.ascontiguousarray(roots[n // 3])):
            rc, want_p, want_y = orc.compute_proof(test_srs_wire, wire, roots, z, literal=False)
            for srs in (srs_lag, srs_plain):
                proof, y = kzg._compute_proof_impl(poly, z, srs, want_y=True)
                assert np.array_equal(proof, want_p) and np.array_equal(y, want_y), (log_n, srs is srs_lag)
            got = list(kzg.compute_proof_stream([(poly, z)] * 3, srs_lag, want_y=True))            # the asynchronous slots take the same path
            assert len(got) == 3 and all(np.array_equal(p_, want_p) and np.array_equal(y_, want_y) for p_, y_ in got)
    raw = bytes(rnd.randrange(256) for _ in range(32 * 2048 - 5))
    blob = k.Blob.from_padded_unchecked(raw)
    kzg = k.KZG.new(); kzg.calculate_and_store_roots_of_unity(len(raw))
    c_lag = kzg.commit_blob(blob, srs_lag); c_plain = kzg.commit_blob(blob, srs_plain)
    assert np.array_equal(c_lag, c_plain)
    assert np.array_equal(kzg.compute_blob_proof(blob, c_lag, srs_lag), kzg.compute_blob_proof(blob, c_plain, srs_plain))
    a = kzg.commit_and_prove_blob(blob, srs_lag); b = kzg.commit_and_prove_blob(blob, srs_plain)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert list(map(bytes, kzg.commit_blob_stream([blob, blob], srs_lag))) == [bytes(c_plain)] * 2
    srs_plain.close(); srs_lag.close()


def test_step_order_and_argument_guards(k, ref_srs):
    lib = k._lib.load(); ctx = k.default_context(); P = k._lib.ptr
    n = 64
    sh = ref_srs.lagrange_shard(n, 16, 16)
    ev = pyref.frs_to_mont(list(range(1, 17)))
    z = pyref.fr_to_mont(5)
    buf8 = np.zeros(8, np.uint64); buf32 = np.zeros(32, np.uint64); y = np.zeros(4, np.uint64)
    INV = k._lib.ERR_INVALID_ARG
    assert lib.kzg_compute_proof_lagrange_partial_y(ctx.handle, 0, P(buf8)) == INV          # nothing in flight
    assert lib.kzg_compute_proof_lagrange_continue(ctx.handle, 0, P(y)) == INV
    assert lib.kzg_compute_proof_lagrange_end(ctx.handle, 0, P(buf32)) == INV
    assert lib.kzg_compute_proof_lagrange_begin(ctx.handle, sh.handle, 16, P(ev), 16, 48, P(z), 0) == k._lib.ERR_INVALID_INPUT_LENGTH   # n not a power of two
    assert lib.kzg_compute_proof_lagrange_begin(ctx.handle, sh.handle, 60, P(ev), 16, n, P(z), 0) == INV      # slice past the domain
    assert lib.kzg_compute_proof_lagrange_begin(ctx.handle, sh.handle, 0, P(np.zeros((17, 4), np.uint64)), 17, n, P(z), 0) == k._lib.ERR_SRS_CAPACITY_EXCEEDED
    assert lib.kzg_compute_proof_lagrange_begin(ctx.handle, sh.handle, 16, P(ev), 16, n, P(z), 9) == INV      # no such slot
    assert lib.kzg_compute_proof_lagrange_begin(ctx.handle, sh.handle, 16, P(ev), 16, n, P(z), 1) == 0
    assert lib.kzg_compute_proof_lagrange_begin(ctx.handle, sh.handle, 16, P(ev), 16, n, P(z), 1) == INV      # the slot is busy
    assert lib.kzg_compute_proof_lagrange_continue(ctx.handle, 1, P(y)) == INV                                  # y before the partial was collected
    assert lib.kzg_compute_proof_lagrange_partial_y(ctx.handle, 1, P(buf8)) == 0
    assert lib.kzg_compute_proof_lagrange_end(ctx.handle, 1, P(buf32)) == INV
    assert lib.kzg_compute_proof_lagrange_abort(ctx.handle, 1) == 0
    assert lib.kzg_compute_proof_lagrange_begin(ctx.handle, sh.handle, 16, P(ev), 16, n, P(z), 1) == 0        # free again
    assert lib.kzg_compute_proof_lagrange_abort(ctx.handle, 1) == 0
    # a fold whose parts do not cover the domain point reports it instead of returning a wrong proof
    rc, roots = orc.calculate_roots_of_unity(n * 32)
    zon = np.ascontiguousarray(roots[3])                                                     # index 3 is outside [16, 32)
    assert lib.kzg_compute_proof_lagrange_begin(ctx.handle, sh.handle, 16, P(ev), 16, n, P(zon), 0) == 0
    assert lib.kzg_compute_proof_lagrange_partial_y(ctx.handle, 0, P(buf8)) == 0 and not buf8.any()
    assert lib.kzg_compute_proof_lagrange_continue(ctx.handle, 0, P(y)) == 0
    assert lib.kzg_compute_proof_lagrange_end(ctx.handle, 0, P(buf32)) == 0
    out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
    assert lib.kzg_lagrange_fold_proof(P(buf32), 1, n, P(zon), P(out), C.byref(inf)) == k._lib.ERR_ROOT_NOT_FOUND
    sh.close()


def test_python_mirror_one_rank(k, ref_srs, test_srs_wire):
    """sharding.ShardedKzgLagrange with world = 1 (no collective): the whole path through the Python host."""
    from rust_kzg_bn254_amd.sharding import ShardedKzgLagrange
    n = 512
    rnd = random.Random(5)
    wire = pyref.frs_to_mont([rnd.randrange(R_) for _ in range(n)])
    sk = ShardedKzgLagrange.from_monomial(k.default_context(), ref_srs, n, gather_device=None)
    rc, roots = orc.calculate_roots_of_unity(n * 32)
    rc, want_c = orc.commit_eval_form(test_srs_wire, wire, literal=False)
    assert np.array_equal(sk.commit_eval_form(k.PolynomialEvalForm(wire)), want_c)
    for z in (pyref.fr_to_mont(123456789), np.ascontiguousarray(roots[77])):
        rc, want_p, want_y = orc.compute_proof(test_srs_wire, wire, roots, z, literal=False)
        proof, y = sk.compute_proof(wire, z, want_y=True)
        assert np.array_equal(proof, want_p) and np.array_equal(y, want_y)
    sk.srs.close()


# ---- full size: 2^20 evaluations, 8 uneven shards, big-integer closed forms ------------------------------------------------------------
LOG_N = 20
N = 1 << LOG_N


def test_config4_2_20_eight_uneven_lagrange_shards(k):
    """BASELINE config 4 at full size on the known-tau SRS: commitment == f^(tau) G1, proof == ((f^(tau) - y) / (tau - z)) G1, y == f^(z)
    by big-integer barycentric evaluation (tests/test_gpu_config4.py's expectations) -- through 8 uneven evaluation-index shards (one
    of a single element, one empty) of the 2^20-point Lagrange basis; off the domain and at a domain point owned by the one-element shard."""
    from test_gpu_config4 import Domain, expect_point, proof_scalar, ints_to_mont
    dom = Domain(LOG_N)
    rnd = random.Random(0xC0F4)
    evals = [rnd.randrange(R_) for _ in range(N)]
    wire = ints_to_mont(evals)
    ftau = dom.evaluate(evals, TAU)
    z = rnd.randrange(R_)
    y = dom.evaluate(evals, z)
    srs = k.SRS.generate(TAU, N)
    edges = [0, 100_000, 100_001, 300_000, 300_000, 524_288 + 70_000, 800_000, 1_000_000, N]
    bounds = pairs(edges)
    full = srs.lagrange(N)
    shards = [full.slice(lo, hi - lo) for lo, hi in bounds]
    full.close()
    c, proof, yy = sharded(k, shards, bounds, N, wire, pyref.fr_to_mont(z), slots=True)
    assert pyref.point_from_wire(c) == expect_point(ftau)
    assert pyref.fr_from_mont(yy) == y
    assert pyref.point_from_wire(proof) == expect_point(proof_scalar(ftau, y, z))
    for m in (100_000, 777_777):                              # the one-element shard owns w^100000
        c, proof, yy = sharded(k, shards, bounds, N, wire, pyref.fr_to_mont(dom.roots[m]), slots=True)
        assert pyref.fr_from_mont(yy) == evals[m]
        assert pyref.point_from_wire(proof) == expect_point(proof_scalar(ftau, evals[m], dom.roots[m]))
    # the one-GPU call over the cached basis, same expectations
    srs.cache_lagrange(N)
    kz = k.KZG.new(); kz.calculate_and_store_roots_of_unity(N * 32)
    p1, y1 = kz._compute_proof_impl(k.PolynomialEvalForm(wire), pyref.fr_to_mont(z), srs, want_y=True)
    assert pyref.fr_from_mont(y1) == y and pyref.point_from_wire(p1) == expect_point(proof_scalar(ftau, y, z))
    for s in shards:
        s.close()
    srs.close()


def test_multi_handle_over_lagrange_shards(k, test_srs_wire):
    """kzg_multi_* (one process, several device contexts -- three on this one GPU) after kzg_multi_cache_lagrange(n): eval-form commitments
    and proofs of exactly n evaluations go through the evaluation-index shards (every context reads its own slice of the caller's buffer),
    other lengths keep the replicated path; all against the oracle."""
    from rust_kzg_bn254_amd.sharding import MultiKzg
    mk = MultiKzg([0, 0, 0])
    mk.srs_upload(test_srs_wire[:2048])
    rnd = random.Random(31)
    mk.cache_lagrange(512)
    for n in (512, 256):                                   # 512: Lagrange shards; 256: the replicated path (no basis cached for it)
        wire = pyref.frs_to_mont([rnd.randrange(R_) for _ in range(n)])
        rc, roots = orc.calculate_roots_of_unity(n * 32)
        rc, want_c = orc.commit_eval_form(test_srs_wire, wire, literal=False)
        assert np.array_equal(mk.commit_eval_form(wire), want_c)
        for z in (pyref.fr_to_mont(rnd.randrange(R_)), np.ascontiguousarray(roots[0]), np.ascontiguousarray(roots[n // 3 + 1]), np.ascontiguousarray(roots[n - 1])):
            rc, want_p, want_y = orc.compute_proof(test_srs_wire, wire, roots, z, literal=False)
            proof, y = mk.compute_proof(wire, z)
            assert np.array_equal(proof, want_p) and np.array_equal(y, want_y), n
    with pytest.raises(ValueError):
        mk.cache_lagrange(100)                             # "length provided is not a power of 2"
    with pytest.raises(ValueError):
        mk.cache_lagrange(4096)                            # more points than the SRS holds
    mk.close()


def test_commit_and_prove_stream_one_rank(k, ref_srs, test_srs_wire):
    """sharding.ShardedKzgLagrange.commit_and_prove_stream (two blobs in flight over the four slots; commitment and proof of a blob from ONE
    upload: kzg_commit_and_prove_lagrange_begin) from host slices and from resident device buffers, on and off the domain, against the oracle;
    a consumer that stops early leaves no slot in flight."""
    import torch
    from rust_kzg_bn254_amd.sharding import ShardedKzgLagrange
    n = 1024
    rnd = random.Random(91)
    rc, roots = orc.calculate_roots_of_unity(n * 32)
    sk = ShardedKzgLagrange.from_monomial(k.default_context(), ref_srs, n, gather_device=None)
    blobs, wants = [], []
    for j in range(7):
        wire = pyref.frs_to_mont([rnd.randrange(R_) for _ in range(n)])
        z = np.ascontiguousarray(roots[rnd.randrange(n)]) if j % 3 == 1 else pyref.fr_to_mont(rnd.randrange(R_))
        rc, want_c = orc.commit_eval_form(test_srs_wire, wire, literal=False)
        rc, want_p, want_y = orc.compute_proof(test_srs_wire, wire, roots, z, literal=False)
        blobs.append((wire, z)); wants.append((want_c, want_p, want_y))
    for depth, grp in ((2, False), (1, False), (4, True), (3, None), (1, True)):       # two slots per blob / grouped launches (one slot per blob)
        got = list(sk.commit_and_prove_stream(blobs, depth=depth, grouped=grp))
        assert len(got) == 7
        for (c, p, y), (wc, wp, wy) in zip(got, wants):
            assert np.array_equal(c, wc) and np.array_equal(p, wp) and np.array_equal(y, wy)
    dev = [torch.from_numpy(w.view(np.int64)).cuda() for w, _ in blobs]
    torch.cuda.synchronize()
    got = list(sk.commit_and_prove_stream([(d.data_ptr(), z) for d, (_, z) in zip(dev, blobs)], resident=True))
    assert all(np.array_equal(c, wc) and np.array_equal(p, wp) and np.array_equal(y, wy) for (c, p, y), (wc, wp, wy) in zip(got, wants))
    # grouped launches through the C-ABI: refused where two scalar sets do not fit one launch, and kzg_compute_proof_lagrange_end refuses a grouped slot
    lib = k._lib.load(); ctx = k.default_context(); P = k._lib.ptr
    w0, z0 = blobs[0]
    assert lib.kzg_commit_and_prove_lagrange_begin(ctx.handle, sk.srs.handle, 0, P(w0), n, n, P(z0), 2, 2) == 0
    yp = np.zeros(8, np.uint64); yy = np.zeros(4, np.uint64); part = np.zeros(32, np.uint64); cpart = np.zeros(16, np.uint64)
    assert lib.kzg_compute_proof_lagrange_partial_y(ctx.handle, 2, P(yp)) == 0 and lib.kzg_lagrange_fold_y(P(yp), 1, n, P(z0), P(yy)) == 0
    assert lib.kzg_compute_proof_lagrange_continue(ctx.handle, 2, P(yy)) == 0
    assert lib.kzg_compute_proof_lagrange_end(ctx.handle, 2, P(part)) == k._lib.ERR_INVALID_ARG          # a grouped slot: both sums belong together
    assert lib.kzg_commit_and_prove_lagrange_end(ctx.handle, 2, P(cpart), P(part)) == 0
    from rust_kzg_bn254_amd.sharding import fold_partials
    assert np.array_equal(fold_partials(cpart.reshape(1, 16)), wants[0][0])
    gen = sk.commit_and_prove_stream(blobs)
    first = next(gen); next(gen)
    gen.close()                                                         # blobs still in flight
    assert np.array_equal(first[0], wants[0][0])
    proof, y = sk.compute_proof(blobs[3][0], blobs[3][1], want_y=True)   # every slot is free again: slot 0 serves the one-call form
    assert np.array_equal(proof, wants[3][1]) and np.array_equal(y, wants[3][2])
    sk.srs.close()


def test_randomised_soak_of_the_lagrange_shards():
    """tools/soak_lagrange.py for four seconds with a fixed seed: random domain sizes 2^0 .. 2^13, random shard edges (empty, one element, next to the
    evaluation point), z on and off the domain, the four-step calls / two-slot / grouped launches / the Python stream -- bit-identical to the one-GPU calls."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SOAK_SECONDS="4", SOAK_SEED="20261004")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_lagrange.py")], env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0 and "soak ok" in r.stdout, r.stdout[-400:] + r.stderr[-1200:]


def test_large_host_buffer_commitments_in_two_parts(k):
    """One call at a time from HOST buffers, 2^19 elements: kzg_commit_coeff_form, and over a cached Lagrange basis kzg_commit_eval_form /
    kzg_commit_blob, go in TWO parts on two slots (the second part's upload hidden behind the first part's kernels; capi.hip msm_srs_common).
    Coefficient form against sum_i c_i tau^i G1 by big integers; the eval / blob forms against the SAME calls on an SRS handle without the
    cached basis (IFFT + monomial MSM in one part: an independent path); a ragged blob whose tail falls into the second part."""
    n = 1 << 19
    srs_plain = k.SRS.generate(TAU, n)
    srs_lag = k.SRS.generate(TAU, n)
    srs_lag.cache_lagrange(n)
    rnd = random.Random(519)
    vals = [rnd.randrange(R_) for _ in range(n)]
    wire = pyref.frs_to_mont(vals)
    kz = k.KZG.new()
    acc, tp = 0, 1
    for v in vals:
        acc = (acc + v * tp) % R_
        tp = tp * TAU % R_
    c = kz.commit_coeff_form(k.PolynomialCoeffForm(wire), srs_plain)
    assert pyref.point_from_wire(c) == pyref.ec_mul(acc, (1, 2))
    e_lag = kz.commit_eval_form(k.PolynomialEvalForm(wire), srs_lag)
    e_plain = kz.commit_eval_form(k.PolynomialEvalForm(wire), srs_plain)
    assert np.array_equal(e_lag, e_plain)
    for n_bytes in (32 * n, 32 * ((1 << 18) + 12345) - 9, 32 * ((1 << 18) + 1)):
        raw = bytes(rnd.getrandbits(8) for _ in range(1024)) * (n_bytes // 1024 + 1)
        blob = k.Blob.from_padded_unchecked(raw[:n_bytes])
        assert np.array_equal(kz.commit_blob(blob, srs_lag), kz.commit_blob(blob, srs_plain)), n_bytes
    # slot 0 / 1 busy: the same call falls back to one part and still agrees
    lib = k._lib.load(); ctx = k.default_context(); P = k._lib.ptr
    assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs_plain.handle, 0, P(wire), 4096, 1) == 0
    out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
    assert lib.kzg_commit_coeff_form(ctx.handle, srs_plain.handle, P(wire), n, P(out), C.byref(inf)) == 0 and np.array_equal(out, c)
    assert lib.kzg_msm_g1_srs_end(ctx.handle, 1, P(out), C.byref(inf), None) == 0
    srs_plain.close(); srs_lag.close()
