"""world_size-2 gloo test (CPU) of the multi-GPU path's host logic: shard bounds, the all-gather of XYZZ
partials over torch.distributed, and the product's host fold (kzg_g1_fold_partials).  There is no GPU here, so
the per-rank partial sum is produced by the oracle *inside this test* as a stand-in for the HIP MSM; what is under
test is everything around it (the same code bench.py runs at N > 1)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle as orc
    import pyref
    from rust_kzg_bn254_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pts = []
    for i, line in enumerate(open(os.path.join(HERE, "golden", "srs.g1.points.string"))):
        if i >= n:
            break
        x, y = line.strip().split(",")
        pts.append((int(x), int(y)))
    srs = pyref.points_to_wire(pts)
    rng = np.random.default_rng(7)
    vals = [int.from_bytes(rng.bytes(31), "big") for _ in range(n)]
    if n > 6:
        vals[3] = 0
        vals[5] = pyref.R_ - 1
    sc = pyref.frs_to_mont(vals)
    lo, hi = sharding.shard_bounds(n, rank, world)
    aff = orc.msm_pippenger(srs[lo:hi], sc[lo:hi], threads=1)          # stand-in for kzg_msm_g1_srs_partial_device
    part = np.zeros(16, np.uint64)
    if aff.any():
        part[:8] = aff
        part[8:12] = pyref.fq_to_mont(1)
        part[12:16] = pyref.fq_to_mont(1)
    gathered = sharding.gather_partials(part, world, device=None)
    got = sharding.fold_partials(gathered)
    want = orc.msm_pippenger(srs, sc, threads=1)
    ok = bool(np.array_equal(got, want))

    # the pipelined stream (ShardedMsm.commit_stream, world > 1 branch: exchange started after MSM k, collected one step later)
    # with the two GPU calls replaced by oracle stand-ins; the ordering / drain logic and the collectives are the real ones
    def to_partial(aff):
        part = np.zeros(16, np.uint64)
        if aff.any():
            part[:8] = aff
            part[8:12] = pyref.fq_to_mont(1)
            part[12:16] = pyref.fq_to_mont(1)
        return part

    sets = [pyref.frs_to_mont([(v * (j + 2) + j) % pyref.R_ for v in vals]) for j in range(7)]

    class Fake(sharding.ShardedMsm):
        def begin(self, srs_shard, ptr, slot):
            assert slot not in self.inflight
            self.inflight[slot] = ptr

        def _end_partial(self, slot):
            j = self.inflight.pop(slot)
            return to_partial(orc.msm_pippenger(srs[lo:hi], sets[j][lo:hi], threads=1))

    fake = Fake(None, n, rank, world, gather_device=None)
    fake.inflight = {}
    wants = [orc.msm_pippenger(srs, sets[j], threads=1) for j in range(len(sets))]
    for depth, bucket in ((None, None), (2, 1), (3, 2), (1, 3), (4, 8)):       # default: depth 3, eight steps per exchange (one flush of 7 here)
        outs = list(fake.commit_stream(None, range(len(sets)), depth=depth, bucket=bucket))
        ok = ok and len(outs) == len(sets) and all(np.array_equal(o, wants[j]) for j, o in enumerate(outs)) and not fake.inflight
    ok = ok and list(fake.commit_stream(None, [])) == [] and not fake.inflight

    # the consumer stops early on every rank while an exchange is in flight (the clean-up must wait for it at any
    # world size), then runs a second stream over the same gatherer: results must not be mixed up with the abandoned exchange
    gen = fake.commit_stream(None, range(len(sets)), depth=2, bucket=2)
    first = next(gen)
    ok = ok and np.array_equal(first, wants[0])
    gen.close()
    ok = ok and not fake.inflight and not fake._gatherer.busy
    outs = list(fake.commit_stream(None, range(len(sets)), depth=2, bucket=2))
    ok = ok and len(outs) == len(sets) and all(np.array_equal(o, wants[j]) for j, o in enumerate(outs))

    # a mid-stream failure on ONE rank (begin() of step 3 raises on rank 1 only): every rank must raise ShardError for step 3
    # after yielding steps 0..2, nobody may hang in a collective, and the next stream must work
    class Failing(Fake):
        def begin(self, srs_shard, ptr, slot):
            if ptr == 3 and self.rank == 1:
                raise ValueError("injected failure")
            super().begin(srs_shard, ptr, slot)

    for depth, bucket in ((3, 8), (2, 1), (1, 2)):
        bad = Failing(None, n, rank, world, gather_device=None)
        bad.inflight = {}
        outs, err = [], None
        try:
            for o in bad.commit_stream(None, range(len(sets)), depth=depth, bucket=bucket):
                outs.append(o)
        except sharding.ShardError as e:
            err = e
        ok = ok and err is not None and err.step == 3 and err.ranks == [1] and not bad.inflight
        ok = ok and len(outs) <= 3 and all(np.array_equal(o, wants[j]) for j, o in enumerate(outs))
        ok = ok and (("injected failure" in str(err)) == (rank == 1))
        outs = list(bad.commit_stream(None, [0, 1, 2], depth=depth, bucket=bucket))
        ok = ok and len(outs) == 3 and all(np.array_equal(o, wants[j]) for j, o in enumerate(outs))

    # several steps per launch (group > 1: begin_group / _end_partials; the batched table mode on a GPU): same results in order at
    # group sizes that do and do not divide the bucket and the stream, and a failing group launch poisons every step of its group
    class FakeG(Fake):
        def begin_group(self, srs_shard, ptrs, slot):
            assert slot not in self.inflight and len(ptrs) > 1
            if 3 in ptrs and self.rank == 1 and getattr(self, "fail", False):
                raise ValueError("injected failure")
            self.inflight[slot] = list(ptrs)

        def _end_partials(self, slot, count):
            if count == 1:
                return [self._end_partial(slot)]
            js = self.inflight.pop(slot)
            assert len(js) == count
            return [to_partial(orc.msm_pippenger(srs[lo:hi], sets[j][lo:hi], threads=1)) for j in js]

    fg = FakeG(None, n, rank, world, gather_device=None)
    fg.inflight = {}
    for depth, bucket, group in ((3, 8, 4), (2, 8, 3), (2, 4, 2), (1, 2, 2), (3, 3, 2), (2, 8, 8)):
        outs = list(fg.commit_stream(None, range(len(sets)), depth=depth, bucket=bucket, group=group))
        ok = ok and len(outs) == len(sets) and all(np.array_equal(o, wants[j]) for j, o in enumerate(outs)) and not fg.inflight
    fg.fail = True
    outs, err = [], None
    try:
        for o in fg.commit_stream(None, range(len(sets)), depth=2, bucket=4, group=2):
            outs.append(o)
    except sharding.ShardError as e:
        err = e
    ok = ok and err is not None and err.step == 2 and err.ranks == [1] and not fg.inflight     # steps 2 and 3 share the failed launch
    ok = ok and len(outs) <= 2 and all(np.array_equal(o, wants[j]) for j, o in enumerate(outs))
    fg.fail = False
    outs = list(fg.commit_stream(None, range(len(sets)), depth=2, bucket=4, group=2))
    ok = ok and len(outs) == len(sets) and all(np.array_equal(o, wants[j]) for j, o in enumerate(outs))

    # an end() that fails on one rank is reported the same way
    class FailingEnd(Fake):
        def _end_partial(self, slot):
            j = self.inflight[slot]
            if j == 2 and self.rank == 0:
                self.inflight.pop(slot)
                raise ValueError("injected end failure")
            return super()._end_partial(slot)

    bad = FailingEnd(None, n, rank, world, gather_device=None)
    bad.inflight = {}
    try:
        list(bad.commit_stream(None, range(5), depth=2, bucket=2))
        ok = False
    except sharding.ShardError as e:
        ok = ok and e.step == 2 and e.ranks == [0]
    q.put((rank, ok, gathered.shape))
    dist.destroy_process_group()


def _run_msm_world(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    for _, ok, shape in res:
        assert ok and tuple(shape) == (world, 16)


@pytest.mark.parametrize("n", [64, 5, 1])
def test_sharded_msm_allgather_and_fold_world2(n):
    _run_msm_world(2, n)


def test_sharded_msm_allgather_and_fold_world8():
    """The rank count of the driver's SCALE run (N = 8), which a one-GPU box cannot rehearse with the GPU (at most 6 processes may use the
    card there): the same host protocol -- shard bounds, bucketed all-gathers, grouped launches, mid-stream failures on one rank -- over
    gloo with eight ranks and 64 / 20 pairs (20: shards of 2 and 3 pairs)."""
    _run_msm_world(8, 64)
    _run_msm_world(8, 20)


def test_shard_bounds_cover_everything():
    from rust_kzg_bn254_amd import sharding
    for n in (0, 1, 7, 1 << 20):
        for world in (1, 2, 3, 8):
            b = [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))


# ---- round 5: BASELINE config 4 sharded by evaluation index (sharding.ShardedKzgLagrange), world 2 over gloo -----------------------------
# No GPU here: the four device steps of a rank (kzg_compute_proof_lagrange_begin / _partial_y / _continue / _end and the partial
# commitment) are replaced by big-integer stand-ins written in this test; everything around them is the product's own code -- the two
# exchanges, the POISON protocol, and the HOST folds of the library (kzg_lagrange_fold_y, kzg_lagrange_fold_proof, kzg_g1_fold_partials:
# host-only entry points, callable without a device).  Expectations: the reference's golden proofs (kzg.proof.eq.input, z on the domain)
# over the reference's own Lagrange points (lagrangeG1SRS.txt), and the oracle for z off the domain.
def _lag_worker(rank, world, port, q):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import ctypes as C
    import torch.distributed as dist
    import oracle as orc
    import pyref
    from pyref import R_
    from rust_kzg_bn254_amd import _lib, sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    golden = os.path.join(HERE, "golden")
    lag = [tuple(int(v) for v in ln.strip().split(",")) for ln in open(os.path.join(golden, "lagrangeG1SRS.txt")) if ln.strip()]
    n = len(lag)
    raw = open(os.path.join(golden, "gettysburg.txt"), encoding="utf-8").read().encode("utf-8")
    evals = pyref.to_fr_array(pyref.pad_payload(raw))
    evals += [0] * (n - len(evals))
    wire = pyref.frs_to_mont(evals)
    w = pyref.root_of_unity(n.bit_length() - 1)
    roots = [pow(w, i, R_) for i in range(n)]
    real = _lib.load()
    state = {}
    commits = {}

    def words(ptr, count):
        return np.ctypeslib.as_array(ptr, shape=(count,))

    def xyzz(pt):
        out = np.zeros(16, np.uint64)
        if pt is not None:
            out[:8] = pyref.point_to_wire(pt)
            out[8:12] = pyref.fq_to_mont(1)
            out[12:16] = pyref.fq_to_mont(1)
        return out

    class FakeLib:
        """the device steps as big-integer arithmetic; every other symbol is the real library's"""
        fail_begin_on = None

        def __getattr__(self, name):
            return getattr(real, name)

        def kzg_commit_eval_form_lagrange_partial(self, ctx, shard, ev_ptr, length, out_ptr):
            lo = shard
            ev = pyref.frs_from_mont(words(ev_ptr, 4 * length).reshape(-1, 4)) if length else []
            words(out_ptr, 16)[:] = xyzz(pyref.msm(lag[lo:lo + length], ev))
            return 0

        def kzg_compute_proof_lagrange_begin(self, ctx, shard, lo, ev_ptr, length, n_, z_ptr, slot):
            if self.fail_begin_on == rank:
                return _lib.ERR_SRS_CAPACITY_EXCEEDED
            ev = pyref.frs_from_mont(words(ev_ptr, 4 * length).reshape(-1, 4)) if length else []
            state[slot] = {"lo": lo, "ev": ev, "z": pyref.fr_from_mont(words(z_ptr, 4).copy())}
            return 0

        def kzg_commit_and_prove_lagrange_begin(self, ctx, shard, lo, ev_ptr, length, n_, z_ptr, cslot, pslot):
            rc = self.kzg_compute_proof_lagrange_begin(ctx, shard, lo, ev_ptr, length, n_, z_ptr, pslot)
            if rc == 0 and length:
                commits[cslot] = xyzz(pyref.msm(lag[lo:lo + length], state[pslot]["ev"]))
            return rc

        def kzg_srs_has_bit_tables(self, shard, build):
            return 1

        def kzg_commit_and_prove_lagrange_end(self, ctx, slot, out_commit, out_part):      # grouped launches: both partial sums of the slot
            words(out_commit, 16)[:] = commits.pop(slot) if slot in commits else 0
            return self.kzg_compute_proof_lagrange_end(ctx, slot, out_part)

        def kzg_msm_g1_srs_end(self, ctx, slot, out_xy, out_inf, out_xyzz):
            if slot not in commits:
                return _lib.ERR_INVALID_ARG
            words(out_xyzz, 16)[:] = commits.pop(slot)
            return 0

        def kzg_compute_proof_lagrange_partial_y(self, ctx, slot, out_ptr):
            st = state[slot]
            z, lo, ev = st["z"], st["lo"], st["ev"]
            out = words(out_ptr, 8)
            out[:] = 0
            if z in roots:
                m = roots.index(z)
                if lo <= m < lo + len(ev):
                    out[4:8] = pyref.fr_to_mont(ev[m - lo])
            else:
                s_ = sum(f * roots[lo + i] % R_ * pow(z - roots[lo + i], -1, R_) for i, f in enumerate(ev)) % R_
                out[:4] = pyref.fr_to_mont(s_)
            return 0

        fail_continue_on = None          # (rank, number of the continue call that fails)
        continues = 0

        def kzg_compute_proof_lagrange_continue(self, ctx, slot, y_ptr):
            FakeLib.continues += 1
            if self.fail_continue_on == (rank, FakeLib.continues):
                return _lib.ERR_DEVICE
            state[slot]["y"] = pyref.fr_from_mont(words(y_ptr, 4).copy())
            return 0

        def kzg_compute_proof_lagrange_end(self, ctx, slot, out_ptr):
            st = state.pop(slot)
            z, lo, ev, y = st["z"], st["lo"], st["ev"], st["y"]
            m = roots.index(z) if z in roots else None
            qs = [0 if lo + i == m else (f - y) * pow(roots[lo + i] - z, -1, R_) % R_ for i, f in enumerate(ev)]
            out = words(out_ptr, 32)
            out[:] = 0
            out[:16] = xyzz(pyref.msm(lag[lo:lo + len(ev)], qs))
            if m is not None:
                out[16:20] = pyref.fr_to_mont(sum(qv * roots[lo + i] for i, qv in enumerate(qs)) % R_)
                if lo <= m < lo + len(ev):
                    out[20:28] = pyref.point_to_wire(lag[m])
                    out[28] = 1
            return 0

        def kzg_compute_proof_lagrange_abort(self, ctx, slot):
            state.pop(slot, None)
            commits.pop(slot, None)                          # (a grouped slot also holds the blob's commitment)
            return 0

    fake = FakeLib()
    sharding._lib.load = lambda: fake                       # this worker process only

    class Ctx:
        handle = None

        def check_device(self, rc):
            pass

    class Shard:                                            # the stand-in passes the slice's first index where the library takes a handle
        def __init__(self, lo, length):
            self.handle, self._n = lo, length

        def __len__(self):
            return self._n

    bounds = (0, 23) if rank == 0 else (23, n)               # uneven on purpose
    sk = sharding.ShardedKzgLagrange(Ctx(), Shard(bounds[0], bounds[1] - bounds[0]), n, rank, world, gather_device=None, bounds=bounds)
    failed = []

    def check(name, cond):
        if not cond:
            failed.append(name)

    # commitment == the reference's literal form over its own Lagrange points
    want_c = pyref.point_to_wire(pyref.msm(lag, evals))
    check("commitment", np.array_equal(sk.commit_eval_form(wire), want_c))
    # z on the domain: the reference's golden proofs (ten of the forty rows, both owners)
    rows = [ln.strip().split(",") for ln in open(os.path.join(golden, "kzg.proof.eq.input")) if ln.strip()]
    for idx, x, y in rows[::4]:
        proof, yy = sk.compute_proof(wire, pyref.fr_to_mont(roots[int(idx)]), want_y=True)
        check("golden proof %s" % idx, pyref.point_from_wire(proof) == (int(x), int(y)))
        check("y on the domain %s" % idx, pyref.fr_from_mont(yy) == evals[int(idx)])
    # z off the domain: y and the proof against big integers (q(x) = (f(x) - y) / (x - z) committed over the same basis)
    for z in (5, 987654321987654321):
        inv = [pow(z - r_, -1, R_) for r_ in roots]
        yv = sum(f * r_ % R_ * iv for f, r_, iv in zip(evals, roots, inv)) % R_ * (pow(z, n, R_) - 1) % R_ * pow(n, -1, R_) % R_
        want_p = pyref.point_to_wire(pyref.msm(lag, [(f - yv) * (R_ - iv) % R_ for f, iv in zip(evals, inv)]))
        proof, yy = sk.compute_proof(wire, pyref.fr_to_mont(z), want_y=True)
        check("y off the domain %d" % z, pyref.fr_from_mont(yy) == yv)
        check("proof off the domain %d" % z, np.array_equal(proof, want_p))
    # a local failure on ONE rank: both ranks raise ShardError naming it, after the same two collectives; the next proof works
    FakeLib.fail_begin_on = 1
    try:
        sk.compute_proof(wire, pyref.fr_to_mont(5))
        check("ShardError raised", False)
    except sharding.ShardError as e:
        check("ShardError names rank 1", e.ranks == [1] and not state)
    FakeLib.fail_begin_on = None
    idx, x, y = rows[7]
    proof = sk.compute_proof(wire, pyref.fr_to_mont(roots[int(idx)]))
    check("proof after a failed call", pyref.point_from_wire(proof) == (int(x), int(y)))
    # ---- the stream: commitment + proof per blob, two blobs in flight, the same results in order on both ranks ----
    zs = [pyref.fr_to_mont(roots[int(r_[0])]) for r_ in rows[:5]] + [pyref.fr_to_mont(5)]
    outs = list(sk.commit_and_prove_stream([(wire, z_) for z_ in zs], depth=2, grouped=False))
    check("stream length", len(outs) == len(zs))
    for dpt in (4, 3):                                   # grouped launches: one slot per blob, up to four blobs in flight
        outs_g = list(sk.commit_and_prove_stream([(wire, z_) for z_ in zs], depth=dpt))
        check("grouped stream (depth %d) == two-slot stream" % dpt, len(outs_g) == len(outs) and all(np.array_equal(a_[0], b_[0]) and np.array_equal(a_[1], b_[1])
                                                                                                       and np.array_equal(a_[2], b_[2]) for a_, b_ in zip(outs_g, outs)))
        check("grouped stream drained", not state and not commits)
    for j, (c_, p_, y_) in enumerate(outs[:5]):
        check("stream commitment %d" % j, np.array_equal(c_, want_c))
        check("stream golden proof %d" % j, pyref.point_from_wire(p_) == (int(rows[j][1]), int(rows[j][2])))
    check("stream drained", not state and not commits)
    check("stream depth 1", all(np.array_equal(a_[1], b_[1]) for a_, b_ in zip(outs, sk.commit_and_prove_stream([(wire, z_) for z_ in zs], depth=1))))
    # a rank that fails BEFORE the y exchange of blob 2 (its begin), and one that fails AFTER it (its continue): in both cases both ranks
    # raise ShardError naming rank 1 after the same collectives, blobs 0 .. are yielded only when every rank had them, and nothing stays in flight
    for mode, grp in (("begin", False), ("continue", False), ("begin", None), ("continue", None)):
        FakeLib.continues = 0
        outs, err = [], None

        def items():
            for j, z_ in enumerate(zs):
                if mode == "begin" and j == 2:
                    FakeLib.fail_begin_on = 1
                yield wire, z_
        if mode == "continue":
            FakeLib.fail_continue_on = (1, 3)
        try:
            for o_ in sk.commit_and_prove_stream(items(), depth=2 if grp is False else 4, grouped=grp):
                outs.append(o_)
        except sharding.ShardError as e:
            err = e
        FakeLib.fail_begin_on = None
        FakeLib.fail_continue_on = None
        check("stream failure (%s) raised on every rank" % mode, err is not None and err.ranks == [1])
        check("stream failure (%s): what was yielded is right" % mode, len(outs) <= 3 and all(np.array_equal(o_[0], want_c) for o_ in outs))
        check("stream failure (%s): nothing in flight" % mode, not state and not commits)
        check("stream after a failure (%s)" % mode, len(list(sk.commit_and_prove_stream([(wire, zs[0])] * 3))) == 3 and not state and not commits)
    # a wrong-sized slice handed to ONE rank (a host-side error before any C-ABI call): the same lockstep failure, not a hang
    err = None
    try:
        list(sk.commit_and_prove_stream([(wire, zs[0]), (wire[:-1] if rank == 1 else wire, zs[1]), (wire, zs[2])], depth=2, grouped=False))
    except sharding.ShardError as e:
        err = e
    check("wrong-sized slice on rank 1: ShardError on every rank", err is not None and err.ranks == [1] and not state and not commits)
    q.put((rank, failed))
    dist.destroy_process_group()


def test_lagrange_sharded_commit_and_proof_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lag_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, []), (1, [])], res
