"""-m gpu: the threading contract of include/kzg_bn254_mi355x.h ("THREADS"), exercised the way the reference's tests use the
library -- ONE `KZG` + `SRS` shared by every test thread (prover/tests/kzg_test.rs:9-17 `lazy_static`,
primitives/tests/blob_test.rs:83-94 rayon):

* 8 host threads hammering ONE context + ONE SRS with the calls that build tables lazily (batched commitment -> per-bit tables of a
  small SRS, g1_ifft(64..256) -> x3 tables, commit_eval_form with a cached Lagrange basis) mixed with plain commitments and proofs;
* 2 contexts x 4 threads sharing nothing but the GPU;
* 2 contexts sharing ONE SRS (an SRS may be used by every context of its GPU), where the first batched call of either context builds
  the tables the other one then finds.

Every result is compared with the oracle (computed once, serially, before the threads start).  ctypes releases the GIL around each
C call, so the threads really are inside the library together."""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle as orc
import pyref

pytestmark = pytest.mark.gpu
N = 1024                     # polynomial length of the jobs (SRS: the reference's 3000 test points)


@pytest.fixture(scope="module")
def k():
    import rust_kzg_bn254_amd as k
    k.load()
    return k


def rand_scalars(n, seed):
    rng = np.random.default_rng(seed)
    vals = [int.from_bytes(rng.bytes(40), "little") % pyref.R_ for _ in range(n)]
    return pyref.frs_to_mont(vals)


@pytest.fixture(scope="module")
def expected(test_srs_wire):
    """Oracle results of the job mix, computed once."""
    rc, roots = orc.calculate_roots_of_unity(N * 32)
    assert rc == N
    out = {"roots": roots, "coeff": [], "eval": [], "proof": []}
    for j in range(4):
        sc = rand_scalars(N, 7000 + j)
        out["coeff"].append((sc, orc.msm_pippenger(test_srs_wire[:N], sc)))
        rc, want = orc.commit_eval_form(test_srs_wire, sc, literal=False)
        assert rc == 0
        out["eval"].append((sc, want))
        z = pyref.fr_to_mont(1234567 + j) if j % 2 == 0 else roots[17 + j]          # off and on the domain
        rc, wantp, wanty = orc.compute_proof(test_srs_wire, sc, roots, z, literal=False)
        assert rc == 0
        out["proof"].append((sc, z, wantp, wanty))
    for n_f in (64, 128):
        rc, lag = orc.g1_ifft(test_srs_wire, n_f)
        assert rc == 0
        out["ifft%d" % n_f] = lag
    batch = np.ascontiguousarray(np.concatenate([out["coeff"][j][0][:256] for j in range(4)]))
    out["batch"] = (batch, [orc.msm_pippenger(test_srs_wire[:256], batch[256 * j:256 * (j + 1)]) for j in range(4)])
    return out


def _jobs(k, lib, ctx, srs, exp, errors, tag, rounds):
    """The job mix of one thread; mismatches are collected, not raised (threads)."""
    from rust_kzg_bn254_amd import _lib

    def check(name, got, want):
        if not np.array_equal(np.asarray(got), np.asarray(want)):
            errors.append("%s: %s differs from the oracle" % (tag, name))

    o8 = np.zeros(8, np.uint64); o4 = np.zeros(4, np.uint64); inf = C.c_uint8(0)
    own = srs.ctx is ctx                                         # commit_eval_form / compute_proof / g1_ifft / the batched forms need the SRS's own context
    for r in range(rounds):
        j = (r + int(tag.rsplit("t", 1)[1])) % 4
        sc, want = exp["coeff"][j]
        rc = lib.kzg_commit_coeff_form(ctx.handle, srs.handle, _lib.ptr(sc), N, _lib.ptr(o8), C.byref(inf))
        check("commit_coeff_form rc", rc, 0); check("commit_coeff_form", o8, want)
        if own:
            sc, want = exp["eval"][j]
            rc = lib.kzg_commit_eval_form(ctx.handle, srs.handle, _lib.ptr(sc), N, _lib.ptr(o8), C.byref(inf))
            check("commit_eval_form rc", rc, 0); check("commit_eval_form", o8, want)
            sc, z, wantp, wanty = exp["proof"][j]
            rc = lib.kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(sc), N, None, N, _lib.ptr(np.ascontiguousarray(z)), _lib.ptr(o8), C.byref(inf), _lib.ptr(o4))
            check("compute_proof rc", rc, 0); check("proof", o8, wantp); check("y", o4, wanty)
        if own:
            n_f = 64 if r % 2 == 0 else 128
            lag = np.zeros((n_f, 8), np.uint64)
            rc = lib.kzg_g1_ifft(ctx.handle, srs.handle, n_f, _lib.ptr(lag))          # first call builds the x3 tables
            check("g1_ifft rc", rc, 0); check("g1_ifft(%d)" % n_f, lag, exp["ifft%d" % n_f])
            batch, wants = exp["batch"]
            ob = np.zeros((4, 8), np.uint64)
            rc = lib.kzg_commit_coeff_form_batch(ctx.handle, srs.handle, _lib.ptr(batch), 256, 4, _lib.ptr(ob), None)   # first call builds the per-bit tables
            check("commit_batch rc", rc, 0)
            for t in range(4):
                check("commit_batch[%d]" % t, ob[t], wants[t])
            if r == 1:
                rc = lib.kzg_srs_cache_lagrange(ctx.handle, srs.handle, N)            # from now on commit_eval_form is one MSM over it
                check("cache_lagrange rc", rc, 0)
        else:                                                    # another context of the same GPU sharing the SRS: the MSM entries
            sc, want = exp["coeff"][(j + 1) % 4]
            rc = lib.kzg_msm_g1_srs(ctx.handle, srs.handle, 0, _lib.ptr(sc), N, _lib.ptr(o8), C.byref(inf))
            check("msm_g1_srs (shared SRS) rc", rc, 0); check("msm_g1_srs (shared SRS)", o8, want)


def _run(threads):
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a worker thread is stuck inside the library"


def test_eight_threads_share_one_context_and_srs(k, test_srs_wire, expected):
    lib = k.load()
    ctx = k.Context(0)
    srs = k.SRS(test_srs_wire, ctx=ctx)                          # 3000 points: no per-bit tables yet (built by the first batched call)
    assert lib.kzg_srs_has_bit_tables(srs.handle, 0) in (0, 1)
    errors = []
    _run([threading.Thread(target=_jobs, args=(k, lib, ctx, srs, expected, errors, "shared/t%d" % i, 4)) for i in range(8)])
    assert not errors, errors[:5]
    assert lib.kzg_srs_has_bit_tables(srs.handle, 0) == 1        # the batched calls built them, once
    srs.close()


def test_two_contexts_four_threads_each_share_nothing(k, test_srs_wire, expected):
    lib = k.load()
    ctxs = [k.Context(0), k.Context(0)]
    srss = [k.SRS(test_srs_wire, ctx=c) for c in ctxs]
    errors = []
    _run([threading.Thread(target=_jobs, args=(k, lib, ctxs[i % 2], srss[i % 2], expected, errors, "ctx%d/t%d" % (i % 2, i), 3)) for i in range(8)])
    assert not errors, errors[:5]
    for s in srss:
        s.close()


def test_two_contexts_share_one_srs(k, test_srs_wire, expected):
    """Context B runs MSMs over context A's SRS while A's threads make the SRS grow its lazy tables."""
    lib = k.load()
    a, b = k.Context(0), k.Context(0)
    srs = k.SRS(test_srs_wire, ctx=a)
    errors = []
    _run([threading.Thread(target=_jobs, args=(k, lib, a if i % 2 == 0 else b, srs, expected, errors, "%s/t%d" % ("AB"[i % 2], i), 3)) for i in range(6)])
    assert not errors, errors[:5]
    srs.close()


def test_sharded_stream_groups_only_with_bit_tables(k, monkeypatch):
    """ADVICE r3 (sharding.py auto_group): the grouped launch needs the shard's per-bit tables; a shard without them streams one launch
    per step instead of failing every launch."""
    from rust_kzg_bn254_amd.sharding import ShardedMsm
    ctx = k.Context(0)
    tau = 0x1234567
    sh = ShardedMsm(ctx, 1 << 16, 0, 1, gather_device=None)
    srs = k.SRS.generate(tau, 1 << 16, ctx=ctx)
    assert sh.auto_group(srs) >= 1
    monkeypatch.setenv("KZG_NO_NAF", "1")
    small = k.SRS.generate(tau, 1 << 10, ctx=ctx)               # below the size that gets tables at upload, and KZG_NO_NAF forbids the build
    sh2 = ShardedMsm(ctx, 1 << 14, 0, 1, gather_device=None)
    assert sh2.auto_group(small) == 1
    small.close(); srs.close()
