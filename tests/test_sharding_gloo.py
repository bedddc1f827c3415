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


@pytest.mark.parametrize("n", [64, 5, 1])
def test_sharded_msm_allgather_and_fold_world2(n):
    world = 2
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
    assert sorted(r[0] for r in res) == [0, 1]
    for _, ok, shape in res:
        assert ok and tuple(shape) == (2, 16)


def test_shard_bounds_cover_everything():
    from rust_kzg_bn254_amd import sharding
    for n in (0, 1, 7, 1 << 20):
        for world in (1, 2, 3, 8):
            b = [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
