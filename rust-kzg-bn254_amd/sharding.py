"""Multi-GPU sharding of the G1 MSM: one process per GPU, rank g owns the contiguous slice
[g n / G, (g+1) n / G) of the SRS (resident on its GPU) and of the scalars; each rank runs a full Pippenger on
its slice and emits ONE extended-Jacobian partial (16 x u64).  The only exchange is an all-gather of the G
partials (G x 128 B over RCCL / xGMI; RCCL has no EC-add reduction op, so "all-reduce" = all-gather + fold);
every rank then folds them on the host (G - 1 point additions + one inversion, kzg_g1_fold_partials).
"""
import collections
import ctypes as C
import datetime
import os
import time

import numpy as np

from . import _lib

# A rank whose MSM could not be enqueued still takes part in every collective (the other ranks have already issued theirs, or
# will): it sends this partial instead.  All-ones words are >= the field modulus, so no real XYZZ partial looks like it.
POISON = np.full(16, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)


class ShardError(RuntimeError):
    """One or more ranks failed at step `step` of a sharded stream; raised by EVERY rank at that step."""

    def __init__(self, step, ranks, cause=None):
        self.step, self.ranks = step, list(ranks)
        msg = "sharded MSM: rank(s) %s failed at step %d" % (self.ranks, step)
        if cause is not None:
            msg += ": %s" % (cause,)
        super().__init__(msg)


class ExchangeTimeout(RuntimeError):
    """The all-gather of the partials did not complete within the time-out (a peer is gone or hung)."""


def exchange_timeout_s():
    return float(os.environ.get("KZG_EXCHANGE_TIMEOUT_S", "60"))


def shard_bounds(n: int, rank: int, world: int):
    return rank * n // world, (rank + 1) * n // world


def gather_partials(partial, world: int, device=None, force_collective: bool = False):
    """all-gather of one 16 x u64 partial per rank -> (world, 16) uint64.  Uses torch.distributed when world > 1
    (backend nccl = RCCL with CUDA tensors, gloo with CPU tensors)."""
    partial = np.ascontiguousarray(partial, dtype=np.uint64).reshape(16)
    if world == 1 and not force_collective:
        return partial.reshape(1, 16).copy()
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(partial.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    outs = [torch.empty(16, dtype=torch.int64, device=t.device) for _ in range(world)]
    dist.all_gather(outs, t)
    return torch.stack(outs).cpu().numpy().view(np.uint64)


class PartialGatherer:
    """The per-step exchange of the sharded MSM with everything preallocated: pinned staging on both sides, one device
    input / output tensor, a private non-blocking torch stream (so the copies and the RCCL kernel neither wait for nor block
    the library's MSM streams), `all_gather_into_tensor`.  Measured on one rank: 170 us per step with fresh pageable tensors
    (most of it a synchronous pageable H2D copy) -> see tools/rehearse_rccl_world1.py.  CPU backends (gloo) take the list form."""

    MAX_BUCKET = 8          # partials one exchange can carry (commit_stream buckets several steps into one collective)
    STREAM_PRIORITY = 0     # of the private stream (tools/probe_exchange_prio.py: high priority gains nothing measurable, for this stream or RCCL's)

    def __init__(self, world: int, device="cuda"):
        import sys
        if device is not None and "torch" not in sys.modules and _lib._lib is not None:
            raise RuntimeError("import torch before the first call into libkzg_bn254_mi355x.so (two HIP runtimes in one process: "
                               "INTEGRATION.md section 5)")
        import torch
        self.world, self.device = world, device
        self.torch = torch
        self._count = 0
        self._work = None
        self._busy = False
        if device is not None:
            m = self.MAX_BUCKET * 16
            self.pin_in = torch.empty(m, dtype=torch.int64).pin_memory()
            self.pin_out = torch.empty(world * m, dtype=torch.int64).pin_memory()
            self.dev_in = torch.empty(m, dtype=torch.int64, device=device)
            self.dev_out = torch.empty(world * m, dtype=torch.int64, device=device)
            self.stream = torch.cuda.Stream(device=self.dev_in.device, priority=self.STREAM_PRIORITY)

    def gather(self, partial):
        """One partial per rank, blocking: (world, 16)."""
        self.start(partial)
        return self.finish().reshape(self.world, 16)

    def start(self, partials):
        """Enqueue the exchange of `count` <= MAX_BUCKET partials ((count, 16) or (16,) u64; the same count on every rank) and
        return at once; `finish()` returns the (world, count, 16) result.  One exchange in flight at a time.  On a saturated GPU
        the tiny copies and the RCCL kernel take ~100 us to get through; started right after an MSM and finished one step later,
        that latency never reaches the host loop, and several steps per exchange divide its ~55 us of host-side calls."""
        import torch.distributed as dist
        torch = self.torch
        partials = np.ascontiguousarray(partials, dtype=np.uint64).reshape(-1, 16)
        count = partials.shape[0]
        if count < 1 or count > self.MAX_BUCKET:
            raise ValueError("1 .. %d partials per exchange" % self.MAX_BUCKET)
        if self._busy:
            raise RuntimeError("PartialGatherer: finish() the exchange in flight before starting another")
        self._count = count
        self._busy = True
        if self.device is None:
            t = torch.from_numpy(partials.view(np.int64).reshape(-1).copy())
            self._cpu_outs = [torch.empty(count * 16, dtype=torch.int64) for _ in range(self.world)]
            self._work = dist.all_gather(self._cpu_outs, t, async_op=True)
            return
        m = count * 16
        self.pin_in.numpy()[:m] = partials.view(np.int64).reshape(-1)
        with torch.cuda.stream(self.stream):
            self.dev_in[:m].copy_(self.pin_in[:m], non_blocking=True)
            dist.all_gather_into_tensor(self.dev_out[:self.world * m], self.dev_in[:m])
            self.pin_out[:self.world * m].copy_(self.dev_out[:self.world * m], non_blocking=True)
            self._done = torch.cuda.Event()
            self._done.record(self.stream)

    def finish(self, timeout_s=None):
        """Result of the exchange in flight, (world, count, 16).  Waits at most `timeout_s` (default KZG_EXCHANGE_TIMEOUT_S = 60 s):
        a peer that died or never issued its collective would otherwise block this rank for ever; ExchangeTimeout then tells the
        caller to exit non-zero (the communicator is unusable after it)."""
        if not self._busy:
            raise RuntimeError("PartialGatherer: no exchange in flight")
        timeout_s = exchange_timeout_s() if timeout_s is None else float(timeout_s)
        if self.device is None:
            try:
                ok = self._work.wait(datetime.timedelta(seconds=timeout_s))
            except RuntimeError as e:                   # gloo raises on time-out / peer loss
                raise ExchangeTimeout("all-gather of the partials failed: %s" % (e,)) from e
            finally:
                self._busy = False
            if ok is False:
                raise ExchangeTimeout("all-gather of the partials timed out after %.0f s" % timeout_s)
            return self.torch.stack(self._cpu_outs).numpy().view(np.uint64).reshape(self.world, self._count, 16)
        deadline = time.monotonic() + timeout_s
        spins = 0
        while not self._done.query():
            spins += 1
            if spins > 2000:                            # ~first 100 us busy, then yield the core
                time.sleep(50e-6)
            if time.monotonic() > deadline:
                self._busy = False
                raise ExchangeTimeout("all-gather of the partials timed out after %.0f s (a peer rank is gone or hung)" % timeout_s)
        self._busy = False
        m = self._count * 16
        return self.pin_out.numpy()[:self.world * m].view(np.uint64).reshape(self.world, self._count, 16).copy()

    @property
    def busy(self):
        return self._busy


def fold_partials(parts):
    """(count, 16) XYZZ partials -> affine wire point (8 u64); host epilogue of the C-ABI."""
    parts = np.ascontiguousarray(parts, dtype=np.uint64).reshape(-1, 16)
    out = np.zeros(8, dtype=np.uint64)
    inf = C.c_uint8(0)
    rc = _lib.load().kzg_g1_fold_partials(_lib.ptr(parts), parts.shape[0], _lib.ptr(out), C.byref(inf))
    if rc != _lib.OK:
        raise ValueError(_lib.status_message(rc))
    return out


class ShardedMsm:
    """MSM of n (scalar, point) pairs sharded over `world` ranks."""

    def __init__(self, ctx, n: int, rank: int = 0, world: int = 1, gather_device="cuda", force_exchange=None):
        self.ctx, self.n, self.rank, self.world = ctx, n, rank, world
        self.lo, self.hi = shard_bounds(n, rank, world)
        self.len = self.hi - self.lo
        self.gather_device = gather_device
        self._gatherer = None
        # one rank normally folds its own result; `force_exchange` (KZG_SHARD_FORCE_EXCHANGE=1) sends it through the collective all the
        # same -- the N > 1 code path on a one-GPU box (needs an initialised process group of one rank)
        if force_exchange is None:
            force_exchange = os.environ.get("KZG_SHARD_FORCE_EXCHANGE") == "1"
        self.exchange = world > 1 or bool(force_exchange)

    def _gather(self, part):
        if self._gatherer is None:
            self._gatherer = PartialGatherer(self.world, self.gather_device)
        return self._gatherer.gather(part)

    def partial_device(self, srs_shard, d_scalars_ptr: int):
        """Partial sum of this rank's slice; `srs_shard` holds points [lo, hi), scalars are in device memory."""
        out = np.zeros(16, dtype=np.uint64)
        rc = _lib.load().kzg_msm_g1_srs_partial_device(self.ctx.handle, srs_shard.handle, 0, C.c_void_p(d_scalars_ptr), self.len,
                                                       _lib.ptr(out))
        self.ctx.check_device(rc)
        if rc != _lib.OK:
            raise ValueError(_lib.status_message(rc))
        return out

    def commit_device(self, srs_shard, d_scalars_ptr: int):
        if not self.exchange:
            out = np.zeros(8, dtype=np.uint64); inf = C.c_uint8(0)
            rc = _lib.load().kzg_msm_g1_srs_device(self.ctx.handle, srs_shard.handle, 0, C.c_void_p(d_scalars_ptr), self.len,
                                                   _lib.ptr(out), C.byref(inf))
            self.ctx.check_device(rc)
            if rc != _lib.OK:
                raise ValueError(_lib.status_message(rc))
            return out
        part = self.partial_device(srs_shard, d_scalars_ptr)
        return fold_partials(self._gather(part))

    # ---- software-pipelined form (kzg_msm_g1_srs_device_begin / kzg_msm_g1_srs_end) -----------------------------------
    def begin(self, srs_shard, d_scalars_ptr: int, slot: int):
        """Enqueue this rank's partial MSM on `slot` (0 / 1) without waiting."""
        rc = _lib.load().kzg_msm_g1_srs_device_begin(self.ctx.handle, srs_shard.handle, 0, C.c_void_p(d_scalars_ptr), self.len, slot)
        self.ctx.check_device(rc)
        if rc != _lib.OK:
            raise ValueError(_lib.status_message(rc))

    def end(self, slot: int):
        """Wait for `slot`; returns the folded commitment (all-gather of the partials + host fold when world > 1)."""
        if not self.exchange:
            out = np.zeros(8, dtype=np.uint64); inf = C.c_uint8(0)
            rc = _lib.load().kzg_msm_g1_srs_end(self.ctx.handle, slot, _lib.ptr(out), C.byref(inf), None)
        else:
            part = np.zeros(16, dtype=np.uint64)
            rc = _lib.load().kzg_msm_g1_srs_end(self.ctx.handle, slot, None, None, _lib.ptr(part))
        self.ctx.check_device(rc)
        if rc != _lib.OK:
            raise ValueError(_lib.status_message(rc))
        if not self.exchange:
            return out
        return fold_partials(self._gather(part))

    def _end_partial(self, slot: int):
        part = np.zeros(16, dtype=np.uint64)
        rc = _lib.load().kzg_msm_g1_srs_end(self.ctx.handle, slot, None, None, _lib.ptr(part))
        self.ctx.check_device(rc)
        if rc != _lib.OK:
            raise ValueError(_lib.status_message(rc))
        return part

    # ---- several steps of a stream in ONE launch (the batched table mode of the MSM engine) --------------------
    def auto_group(self, srs_shard=None):
        """Steps of a stream this rank's library puts into one launch: shard-sized MSMs are bound by dependent-latency chains, and
        `k` of them over the same SRS slice are one MSM problem with k x the work per kernel (k x 2^(c-1) buckets in one bucket array).
        Derived from n // world, so that every rank groups alike (the ranks must issue the same collectives)."""
        if os.environ.get("KZG_SHARD_GROUP"):
            return max(1, int(os.environ["KZG_SHARD_GROUP"]))
        if os.environ.get("KZG_SHARD_GROUP_AUTO", "1") == "0":
            return 1
        per = self.n // max(1, self.world)
        if per >= (1 << 20) or per < (1 << 13):      # (2^19 pairs per rank, two per launch: 0.579 against 0.597 ms per step)
            return 1
        # the batched launch needs the per-bit tables of the shard (absent with KZG_NO_NAF=1, when memory is short, above 2^22 points):
        # without them one launch per step (every grouped launch failed instead).  Ranks that may disagree (memory) must agree
        # on the minimum before streaming -- bench.py all-reduces it.
        if srs_shard is not None and not _lib.load().kzg_srs_has_bit_tables(srs_shard.handle, 1):
            return 1
        cap = int(_lib.load().kzg_msm_batch_capacity(per))             # per = the smallest shard
        return max(1, min(cap, 4))

    def group_depth(self, depth: int, group: int):
        """Launches in flight for `group` steps per launch: two once a launch covers >= 2^19 pairs.  Such launches saturate the chip with
        two in flight (0.62-0.75 ms per launch of 4 x 2^17 pairs at depth 2 and 3 alike)."""
        if group > 1 and self.len * group >= (1 << 19):
            return min(depth, 2)
        return depth

    def begin_group(self, srs_shard, d_scalars_ptrs, slot: int):
        """Enqueue the partial MSMs of several scalar buffers as ONE launch on `slot` (one buffer: the plain begin())."""
        if len(d_scalars_ptrs) == 1:
            return self.begin(srs_shard, d_scalars_ptrs[0], slot)
        arr = (C.c_void_p * len(d_scalars_ptrs))(*[C.c_void_p(int(p)) for p in d_scalars_ptrs])
        rc = _lib.load().kzg_msm_g1_srs_device_begin_batch(self.ctx.handle, srs_shard.handle, 0, arr, self.len, len(d_scalars_ptrs), slot)
        self.ctx.check_device(rc)
        if rc != _lib.OK:
            raise ValueError(_lib.status_message(rc))

    def _end_partials(self, slot: int, count: int):
        """Wait for the launch on `slot`: its `count` XYZZ partials, in order."""
        if count == 1:
            return [self._end_partial(slot)]
        parts = np.zeros((count, 16), dtype=np.uint64)
        rc = _lib.load().kzg_msm_g1_srs_end_batch(self.ctx.handle, slot, count, None, None, _lib.ptr(parts))
        self.ctx.check_device(rc)
        if rc != _lib.OK:
            raise ValueError(_lib.status_message(rc))
        return [parts[j] for j in range(count)]

    def _end_group(self, slot: int, count: int):
        """world == 1: the `count` commitments of the launch on `slot`."""
        if count == 1:
            return [self.end(slot)]
        out = np.zeros((count, 8), dtype=np.uint64)
        rc = _lib.load().kzg_msm_g1_srs_end_batch(self.ctx.handle, slot, count, _lib.ptr(out), None, None)
        self.ctx.check_device(rc)
        if rc != _lib.OK:
            raise ValueError(_lib.status_message(rc))
        return [out[j] for j in range(count)]

    def commit_stream(self, srs_shard, d_scalars_ptrs, depth=None, bucket=None, group=1):
        """Commitments of a stream of scalar buffers (device pointers to this rank's slices) with `depth` MSMs in flight
        (default: 2 for slices of >= 2^20 pairs, which saturate the GPU's integer pipes, else 3 — shard-sized MSMs are bound by
        dependent-latency chains; a fourth in flight gains or loses depending on the stream -> hardware-queue mapping).  MSM k+depth-1 is enqueued before MSM k is waited for.
        With world > 1 the partials of `bucket` consecutive steps travel in ONE all-gather (default 8), started as soon as the
        bucket's last MSM is done and collected while the next MSMs run (fold on the host).  One exchange per step costs ~55 us of
        host-side calls, and on a GPU saturated by MSM kernels its two copies and the RCCL kernel often have not run yet when the
        host comes to collect them: measured over a one-rank communicator, three MSMs in flight (tools/time_exchange_bucket.py),
        2^19 pairs per step 0.640 ms without exchange, 0.765 / 0.723 / 0.704 / 0.676 ms with 1 / 2 / 4 / 8 steps per exchange;
        2^17 pairs 0.249 against 0.269 / - / 0.251 / 0.250 ms.  The bucket size is fixed (not adaptive): every rank must issue
        collectives of the same size, and must see the same number of items.  Yields the commitments in order.
        Failures with world > 1: a rank whose begin() / end() fails keeps issuing its collectives with a POISON partial for that step
        and does no more GPU work; every rank (the failing one included) raises ShardError at that step, after the same number of
        collectives, so nobody is left blocked in an all-gather.  A peer that dies outright shows up as ExchangeTimeout
        (KZG_EXCHANGE_TIMEOUT_S, default 60 s): exit non-zero on it, the communicator cannot be used again.
        (A Python host should `gc.freeze()` or disable the cyclic collector around a stream: with torch imported one full collection
        takes ~5 ms on this thread, 30 shard steps at 8 ranks, and the GPU then needs ~15 steps to regain its clock -- bench.py does.)
        `group` consecutive steps share ONE launch (begin_group / _end_partials; `None`: auto_group(), 1: one launch per step): at
        2^17 pairs per rank four steps per launch take 0.164 ms per step where one launch per step takes 0.22-0.24 (three in flight;
        2^15: 0.152 -> 0.067, 2^16: 0.167 -> 0.095, 2^18 with two per launch: 0.347 -> 0.311; tools/time_shard_group.py).  The
        group size must be the same on every rank; a failure poisons every step of its group."""
        if depth is None:
            depth = 2 if self.len >= (1 << 20) else 3
        depth = max(1, min(int(depth), _lib.NUM_SLOTS))
        if bucket is None:
            bucket = PartialGatherer.MAX_BUCKET
        bucket = max(1, min(int(bucket), PartialGatherer.MAX_BUCKET))
        if group is None:
            group = self.auto_group()
        group = max(1, min(int(group), bucket))
        depth = self.group_depth(depth, group)
        inflight = collections.deque()
        g = None
        if self.exchange:
            if self._gatherer is None:
                self._gatherer = PartialGatherer(self.world, self.gather_device)
            g = self._gatherer
            if g.busy:                                  # left over by a stream that was abandoned: every rank issued it
                g.finish()
        exchanging = False
        pending = []                                   # partials of finished MSMs not yet sent
        sent_first = 0                                 # step index of the first partial of the exchange in flight
        next_step = 0                                  # step index of the next partial to enter `pending`
        failure = None                                 # this rank's first failure (world > 1: reported through the exchange)

        def collect():
            """results of the exchange in flight, in step order; ShardError on EVERY rank if any rank sent POISON"""
            got = g.finish()                           # (world, count, 16)
            outs = []
            for j in range(got.shape[1]):
                bad = [r for r in range(self.world) if np.array_equal(got[r, j], POISON)]
                if bad:
                    raise ShardError(sent_first + j, bad, failure if self.rank in bad else None)
                outs.append(fold_partials(got[:, j, :]))
            return outs

        def retire(flush=False):
            nonlocal exchanging, sent_first, next_step, failure
            slot, count = inflight.popleft()
            if g is None:
                return self._end_group(slot, count)
            if slot is None:
                parts = [POISON] * count
            else:
                try:
                    parts = self._end_partials(slot, count)
                except Exception as e:                  # noqa: BLE001 -- reported to every rank through the exchange
                    failure = failure or e
                    parts = [POISON] * count
            out = []
            if pending and len(pending) + count > bucket:   # (the groups do not divide the bucket: send what is there first)
                if exchanging:
                    exchanging = False
                    out = collect()
                sent_first = next_step - len(pending)
                g.start(np.stack(pending))
                pending.clear()
                exchanging = True
            pending.extend(parts)
            next_step += count
            if len(pending) == bucket or (flush and not inflight):
                if exchanging:
                    exchanging = False
                    out = out + collect()
                sent_first = next_step - len(pending)
                g.start(np.stack(pending))
                pending.clear()
                exchanging = True
            return out

        def launch(k, ptrs):
            slot = k % depth
            if g is None:
                self.begin_group(srs_shard, ptrs, slot) if len(ptrs) > 1 else self.begin(srs_shard, ptrs[0], slot)
            elif failure is not None:
                slot = None                             # after a failure this rank only keeps the collectives matched
            else:
                try:
                    self.begin_group(srs_shard, ptrs, slot) if len(ptrs) > 1 else self.begin(srs_shard, ptrs[0], slot)
                except Exception as e:                  # noqa: BLE001
                    nonlocal_failure(e)
                    slot = None
            inflight.append((slot, len(ptrs)))

        def nonlocal_failure(e):
            nonlocal failure
            failure = failure or e

        try:
            k = 0
            chunk = []
            for ptr in d_scalars_ptrs:
                chunk.append(ptr)
                if len(chunk) < group:
                    continue
                if len(inflight) == depth:
                    yield from retire()
                launch(k, chunk)
                k += 1
                chunk = []
            if chunk:
                if len(inflight) == depth:
                    yield from retire()
                launch(k, chunk)
            while inflight:
                yield from retire(flush=True)
            if exchanging:
                exchanging = False
                yield from collect()
        finally:
            # a failed begin() or a consumer that stops early must not leave slots (or an exchange) in flight
            while inflight:
                slot, count = inflight.popleft()
                if slot is None:
                    continue
                try:
                    self._end_partials(slot, count)
                except Exception:                       # noqa: BLE001 -- draining: the slot is free again either way
                    pass
            if exchanging and g is not None and g.busy:
                try:
                    g.finish()                          # every rank issued this collective: only a wait
                except ExchangeTimeout:
                    pass


class ShardedKzg:
    """`KZG::commit_eval_form` / `KZG::compute_proof` with the MSM sharded over `world` ranks (BASELINE config 4).
    Every rank holds SRS powers [lo, hi) (`srs_shard`), receives the whole polynomial, does the O(n) field work redundantly
    (IFFT / quotient: 32 B per element, cheaper than exchanging it) and commits its slice of the coefficients; the ranks
    all-gather their 128-byte partials and fold them on the host."""

    def __init__(self, ctx, srs_shard, n: int, rank: int = 0, world: int = 1, gather_device="cuda"):
        self.ctx, self.srs, self.n, self.rank, self.world = ctx, srs_shard, n, rank, world
        self.lo, self.hi = shard_bounds(n, rank, world)
        if len(srs_shard) < self.hi - self.lo:
            raise ValueError("SRS shard shorter than this rank's slice")
        self.gather_device = gather_device

    def _finish(self, part):
        return fold_partials(gather_partials(part, self.world, self.gather_device))

    def commit_eval_form(self, polynomial):
        ev = _lib.as_u64(polynomial.evaluations(), 4)
        part = np.zeros(16, dtype=np.uint64)
        rc = _lib.load().kzg_commit_eval_form_partial(self.ctx.handle, self.srs.handle, self.lo, _lib.ptr(ev), len(ev), _lib.ptr(part))
        self.ctx.check_device(rc)
        if rc != _lib.OK:
            raise ValueError(_lib.status_message(rc))
        return self._finish(part)

    def compute_proof(self, polynomial, z_fr, want_y=False):
        ev = _lib.as_u64(polynomial.evaluations(), 4)
        z = np.ascontiguousarray(_lib.as_u64(z_fr, 0).reshape(4))
        part = np.zeros(16, dtype=np.uint64)
        y = np.zeros(4, dtype=np.uint64)
        rc = _lib.load().kzg_compute_proof_partial(self.ctx.handle, self.srs.handle, self.lo, _lib.ptr(ev), len(ev), None, len(ev),
                                                    _lib.ptr(z), _lib.ptr(part), _lib.ptr(y))
        self.ctx.check_device(rc)
        if rc != _lib.OK:
            raise ValueError(_lib.status_message(rc))
        proof = self._finish(part)
        return (proof, y) if want_y else proof


def gather_words(words, world: int, device=None, force_collective: bool = False):
    """all-gather of one fixed-size u64 vector per rank -> (world, len) uint64 (the two small exchanges of a Lagrange-sharded proof)."""
    words = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
    if world == 1 and not force_collective:
        return words.reshape(1, -1).copy()
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(words.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return torch.stack(outs).cpu().numpy().view(np.uint64)


class ShardedKzgLagrange:
    """`KZG::commit_eval_form` / `KZG::compute_proof` sharded by EVALUATION index (BASELINE config 4 without replicated work): rank g
    holds the Lagrange points L_i = g1_ifft(srs)[i] (prover/src/kzg.rs:263-285) and receives the evaluations f_i of i in [lo, hi) only.
        commit:  sum_{i in slice} f_i L_i                    -> one all-gather of the 128-byte partials            (kzg.rs:96-100)
        proof:   S_g = sum f_i w^i / (z - w^i) on the slice   -> all-gather of 64 B -> y                           (helpers.rs:507-532)
                 q_i = (f_i - y) / (w^i - z) on the slice, MSM over the slice -> all-gather of 256 B -> fold      (kzg.rs:151-177, :237-260)
    No rank uploads, transforms or divides the whole polynomial (ShardedKzg above does all three on every rank).
    A rank whose local step fails still joins both exchanges with a POISON payload; every rank then raises ShardError."""

    YPART, PART = 8, 32

    def __init__(self, ctx, lagrange_shard, n: int, rank: int = 0, world: int = 1, gather_device="cuda", bounds=None, force_exchange=False):
        self.ctx, self.srs, self.n, self.rank, self.world = ctx, lagrange_shard, n, rank, world
        self.lo, self.hi = bounds if bounds is not None else shard_bounds(n, rank, world)
        self.len = self.hi - self.lo
        if len(lagrange_shard) < self.len:
            raise ValueError("Lagrange shard shorter than this rank's slice")
        self.gather_device = gather_device
        self.force = bool(force_exchange)

    @classmethod
    def from_monomial(cls, ctx, srs, n: int, rank: int = 0, world: int = 1, **kw):
        """Set-up from a monomial SRS of >= n points resident on this rank's GPU: g1_ifft(n) once, keep the rank's slice."""
        lo, hi = kw.get("bounds") or shard_bounds(n, rank, world)
        return cls(ctx, srs.lagrange_shard(n, lo, hi - lo), n, rank, world, **kw)

    def _gather(self, words):
        return gather_words(words, self.world, self.gather_device, self.force)

    def _check(self, got, step):
        bad = [r for r in range(got.shape[0]) if np.all(got[r] == np.uint64(0xFFFFFFFFFFFFFFFF))]
        if bad:
            raise ShardError(step, bad)

    def _slice(self, polynomial_or_slice):
        ev = polynomial_or_slice.evaluations() if hasattr(polynomial_or_slice, "evaluations") else polynomial_or_slice
        ev = _lib.as_u64(ev, 4).reshape(-1, 4)
        if len(ev) == self.n and self.len != self.n:
            ev = ev[self.lo:self.hi]                  # a whole polynomial was passed: this rank reads its slice of it
        if len(ev) != self.len:
            raise ValueError("expected the %d evaluations of this rank's slice" % self.len)
        return np.ascontiguousarray(ev)

    def commit_eval_form(self, polynomial_or_slice):
        part = np.zeros(16, dtype=np.uint64)
        try:
            ev = self._slice(polynomial_or_slice)
            rc = _lib.load().kzg_commit_eval_form_lagrange_partial(self.ctx.handle, self.srs.handle, _lib.ptr(ev) if self.len else None,
                                                                   self.len, _lib.ptr(part))
            self.ctx.check_device(rc)
            if rc != _lib.OK:
                raise ValueError(_lib.status_message(rc))
        except Exception:                                   # noqa: BLE001 -- reported to every rank through the exchange
            if self.world == 1 and not self.force:
                raise
            part = POISON.copy()
        got = self._gather(part)
        self._check(got, 0)
        return fold_partials(got)

    def compute_proof(self, polynomial_or_slice, z_fr, want_y=False, slot=0):
        lib = _lib.load()
        z = np.ascontiguousarray(_lib.as_u64(z_fr, 0).reshape(4))
        exchanging = self.world > 1 or self.force
        failed = None
        ypart = np.zeros(self.YPART, dtype=np.uint64)
        try:
            ev = self._slice(polynomial_or_slice)
            rc = lib.kzg_compute_proof_lagrange_begin(self.ctx.handle, self.srs.handle, self.lo, _lib.ptr(ev) if self.len else None, self.len,
                                                      self.n, _lib.ptr(z), slot)
            self.ctx.check_device(rc)
            if rc != _lib.OK:
                raise ValueError(_lib.status_message(rc))
            rc = lib.kzg_compute_proof_lagrange_partial_y(self.ctx.handle, slot, _lib.ptr(ypart))
            self.ctx.check_device(rc)
            if rc != _lib.OK:
                raise ValueError(_lib.status_message(rc))
        except Exception as e:                              # noqa: BLE001
            if not exchanging:
                raise
            failed = e
            ypart = np.full(self.YPART, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
        got = self._gather(ypart)
        bad = [r for r in range(got.shape[0]) if np.all(got[r] == np.uint64(0xFFFFFFFFFFFFFFFF))]
        y = np.zeros(4, dtype=np.uint64)
        part = np.zeros(self.PART, dtype=np.uint64)
        if bad:                                             # every rank still issues the second exchange, then raises
            if failed is None:
                lib.kzg_compute_proof_lagrange_abort(self.ctx.handle, slot)      # a healthy rank: gives up its slot, sends a harmless row
            else:
                part[:] = 0xFFFFFFFFFFFFFFFF
        else:
            try:
                rc = lib.kzg_lagrange_fold_y(_lib.ptr(np.ascontiguousarray(got)), got.shape[0], self.n, _lib.ptr(z), _lib.ptr(y))
                if rc != _lib.OK:
                    raise ValueError(_lib.status_message(rc))
                rc = lib.kzg_compute_proof_lagrange_continue(self.ctx.handle, slot, _lib.ptr(y))
                self.ctx.check_device(rc)
                if rc != _lib.OK:
                    raise ValueError(_lib.status_message(rc))
                rc = lib.kzg_compute_proof_lagrange_end(self.ctx.handle, slot, _lib.ptr(part))
                self.ctx.check_device(rc)
                if rc != _lib.OK:
                    raise ValueError(_lib.status_message(rc))
            except Exception as e:                          # noqa: BLE001
                if not exchanging:
                    raise
                failed = e
                lib.kzg_compute_proof_lagrange_abort(self.ctx.handle, slot)
                part[:] = 0xFFFFFFFFFFFFFFFF
        got2 = self._gather(part)
        bad2 = sorted(set(bad) | {r for r in range(got2.shape[0]) if np.all(got2[r] == np.uint64(0xFFFFFFFFFFFFFFFF))})
        if bad2:
            raise ShardError(0, bad2, failed)
        out = np.zeros(8, dtype=np.uint64)
        inf = C.c_uint8(0)
        rc = lib.kzg_lagrange_fold_proof(_lib.ptr(np.ascontiguousarray(got2)), got2.shape[0], self.n, _lib.ptr(z), _lib.ptr(out), C.byref(inf))
        if rc != _lib.OK:
            raise ValueError(_lib.status_message(rc))
        return (out, y) if want_y else out


    # ---- streams of blobs: commitment AND proof of each, several blobs in flight per rank ---------------------------------------------
    def commit_and_prove_stream(self, items, depth=None, resident=False, grouped=None):
        """BASELINE config 4 as a stream: `items` yields (evaluations, z_fr) -- this rank's slice of the evaluations (host array, or with
        `resident=True` a device pointer to it) and the evaluation point; yields (commitment, proof, y) per blob, in order, the same on
        every rank.  Per blob the rank enqueues ONE upload, the inverses + partial barycentric sum (on the context's high-priority stream) and the
        commitment's MSM (`kzg_commit_and_prove_lagrange_begin`), exchanges 64 B -> y, enqueues quotient + MSM, and later collects both
        partial points and exchanges them (128 + 256 B in one all-gather, collected one step after it was started).  `depth` blobs in flight
        (default 2 with two slots per blob, 3 with grouped launches): the MSMs of earlier blobs run while blob t's phase 1 and both exchanges
        happen, so the ~0.1 ms inversion latency and the exchange latencies are hidden (one call at a time they are half of a 2^17-element
        proof, profiles/r05_config4_shards.md).
        `grouped` (None: when every rank can): one slot per blob, the blob's two MSMs as ONE batched launch once the quotient exists (up to
        four blobs in flight) -- at 2^17-element slices 0.63 -> ~0.45 ms per blob.
        A local failure: the rank sends POISON in the collectives that remain in its iteration and every rank raises ShardError."""
        lib = _lib.load()
        exchanging = self.world > 1 or self.force
        # GROUPED launches (one slot per blob, commitment + proof as one batched launch: kzg_commit_and_prove_lagrange_begin with commit_slot ==
        # proof_slot) when every rank's shard has its per-bit tables and room for two scalar sets per launch; else two slots per blob.  The mode
        # decides how many blobs fit in flight, hence the order of the collectives: the ranks agree on it first (one tiny all-gather).
        # (by default only for slices of <= 2^18 elements: beyond, two sets per launch mean narrower windows than a lone MSM's -- measured per blob
        # with / without grouping, one-rank RCCL exchange: 2^17 0.63 / 0.77, 2^18 0.99 / 0.93-0.99, 2^19 1.65 / 1.40, 2^20 2.96 / 2.68 ms)
        small_enough = self.len <= (1 << 18) or grouped is True
        can_group = (grouped is not False and small_enough and self.len > 0 and bool(lib.kzg_srs_has_bit_tables(self.srs.handle, 1))
                     and int(lib.kzg_msm_batch_capacity(self.len)) >= 2)
        if self.len == 0:
            can_group = grouped is not False
        if exchanging:
            votes = gather_words(np.array([1 if can_group else 0], dtype=np.uint64), self.world, self.gather_device, self.force)
            can_group = bool(votes.min())
        if grouped is True and not can_group:
            raise ValueError("grouped launches need per-bit tables and room for two scalar sets on every rank's shard")
        grouped = can_group
        if depth is None:                                    # measured (tools/trace_config4_stream.py, 2^17-element slices): grouped 0.489 / 0.475 / 0.569 ms per blob
            depth = 3 if grouped else 2                      # with 2 / 3 / 4 blobs in flight; two slots per blob: 0.637 with 2
        depth = max(1, min(int(depth), _lib.NUM_SLOTS if grouped else _lib.NUM_SLOTS // 2))
        g_y = g_r = None
        if exchanging:
            g_y = PartialGatherer(self.world, self.gather_device)
            g_r = PartialGatherer(self.world, self.gather_device)
        ONES = np.uint64(0xFFFFFFFFFFFFFFFF)
        state = {"failed": None, "bad": []}

        def guard(fn, *a):
            """run one local step unless this rank (or a peer) has already failed"""
            if state["failed"] is not None or state["bad"]:
                return None
            try:
                rc = fn(*a)
                self.ctx.check_device(rc)
                if rc != _lib.OK:
                    raise ValueError(_lib.status_message(rc))
                return rc
            except Exception as e:                          # noqa: BLE001 -- reported to every rank through the next collective
                if not exchanging:
                    raise
                state["failed"] = e
                return None

        inflight = collections.deque()                      # (commit slot, proof slot, z, y)

        def release(cs, ps):
            """give two slots back whatever state they are in (best effort: a slot with nothing pending answers INVALID_ARG)"""
            sink = np.zeros(16, dtype=np.uint64)
            if cs != ps:
                lib.kzg_msm_g1_srs_end(self.ctx.handle, cs, None, None, _lib.ptr(sink))
            lib.kzg_compute_proof_lagrange_abort(self.ctx.handle, ps)

        pending_result = []                                  # [z, y] of the blob whose result exchange is in flight (at most one)

        def finish_oldest():
            """wait for the two MSMs of the oldest blob and START the exchange of its partial points (collected one step later, so that
            its latency runs beside the next blob's phase 1)"""
            cs, ps, z, y = inflight.popleft()
            cpart = np.zeros(16, dtype=np.uint64)
            ppart = np.zeros(self.PART, dtype=np.uint64)
            if grouped:
                guard(lib.kzg_commit_and_prove_lagrange_end, self.ctx.handle, ps, _lib.ptr(cpart), _lib.ptr(ppart))
            else:
                if self.len:
                    guard(lib.kzg_msm_g1_srs_end, self.ctx.handle, cs, None, None, _lib.ptr(cpart))
                guard(lib.kzg_compute_proof_lagrange_end, self.ctx.handle, ps, _lib.ptr(ppart))
            if state["failed"] is not None:
                release(cs, ps)                                              # the guarded calls were skipped (or one of them failed)
            rows = np.concatenate([cpart, ppart]).reshape(3, 16)            # commitment partial | the proof's 32 words
            if state["failed"] is not None:
                rows = np.full_like(rows, ONES)
            if exchanging:
                g_r.start(rows)
                pending_result[:] = [z, y, None]
            else:
                pending_result[:] = [z, y, rows.reshape(1, 3, 16)]

        def collect_result():
            """the folded (commitment, proof, y) of the exchange in flight; None (and state["bad"]) when a rank sent POISON"""
            z, y, got = pending_result
            pending_result[:] = []
            if got is None:
                got = g_r.finish()
            bad = [r for r in range(got.shape[0]) if np.all(got[r] == ONES)]
            if bad:
                state["bad"] = sorted(set(state["bad"]) | set(bad))
                return None
            got = np.ascontiguousarray(got.reshape(got.shape[0], 48))
            commitment = fold_partials(got[:, :16])
            proof = np.zeros(8, dtype=np.uint64)
            inf = C.c_uint8(0)
            parts = np.ascontiguousarray(got[:, 16:])
            rc = lib.kzg_lagrange_fold_proof(_lib.ptr(parts), parts.shape[0], self.n, _lib.ptr(z), _lib.ptr(proof), C.byref(inf))
            if rc != _lib.OK:
                raise ValueError(_lib.status_message(rc))
            return commitment, proof, y

        def drain_and_raise():
            for cs, ps, _z, _y in inflight:                                  # give the slots back; no more collectives are issued
                release(cs, ps)
            inflight.clear()
            settle()                                                        # exchanges started on every rank before the failure became visible
            raise ShardError(0, state["bad"], state["failed"])

        # The y exchange of blob t is started and collected ONE ITERATION LATER (with two or more blobs in flight): its latency -- ~0.15 ms per
        # blob on a saturated GPU -- then runs beside the next blob's begin and the collection of older results instead of stalling the host.
        # Only with three or more blobs in flight (grouped launches): with two, the blob whose phase 2 waits a whole iteration leaves the GPU short of
        # work (one-rank RCCL rehearsal, ms per blob without / with the lag: 2^17 0.60 / 0.53, 2^18 0.86 / 0.82, 2^19 1.57 / 1.59, 2^20 2.86 / 3.12).
        lag_y = exchanging and depth >= 3
        pending_y = []                                       # [ps, z, y] of the blob whose y exchange is in flight

        def start_y(ps, z, y, ypart):
            rows = np.ascontiguousarray(ypart, dtype=np.uint64).reshape(1, 16)
            if state["failed"] is not None:
                rows = np.full_like(rows, ONES)
            if exchanging:
                g_y.start(rows)
                pending_y[:] = [ps, z, y, None]
            else:
                pending_y[:] = [ps, z, y, rows.reshape(1, 1, 16)]

        def finish_y():
            """fold y of the exchange in flight and enqueue that blob's phase 2; state["bad"] when a rank sent POISON"""
            ps, z, y, got = pending_y
            pending_y[:] = []
            if got is None:
                got = g_y.finish()
            bad = [r for r in range(got.shape[0]) if np.all(got[r] == ONES)]
            if bad:
                state["bad"] = sorted(set(state["bad"]) | set(bad))
                return
            yp = np.ascontiguousarray(got.reshape(got.shape[0], 16)[:, :8])
            rc = lib.kzg_lagrange_fold_y(_lib.ptr(yp), yp.shape[0], self.n, _lib.ptr(z), _lib.ptr(y))
            if rc != _lib.OK:
                raise ValueError(_lib.status_message(rc))
            # a failure from here on is local knowledge until the NEXT collective of the common schedule: this rank walks on through that
            # schedule without touching the GPU and sends POISON there
            guard(lib.kzg_compute_proof_lagrange_continue, self.ctx.handle, ps, _lib.ptr(y))

        def settle():
            """wait for whatever exchange this rank has started (every rank started the same ones)"""
            for g, pend in ((g_y, pending_y), (g_r, pending_result)):
                if pend and pend[-1 if pend is pending_y else 2] is None and g is not None and g.busy:
                    try:
                        g.finish()
                    except ExchangeTimeout:
                        pass
                pend[:] = []

        t = 0
        try:
            for ev, z_fr in items:
                z = np.ascontiguousarray(_lib.as_u64(z_fr, 0).reshape(4))
                if grouped:
                    cs = ps = t % depth
                else:
                    cs, ps = (2 * t) % (2 * depth), (2 * t + 1) % (2 * depth)
                if pending_result:
                    out = collect_result()
                    if state["bad"]:
                        drain_and_raise()
                    yield out
                if len(inflight) == depth:
                    finish_oldest()
                if resident:
                    guard(lib.kzg_commit_and_prove_lagrange_begin_device, self.ctx.handle, self.srs.handle, self.lo, C.c_void_p(int(ev)) if self.len else None,
                          self.len, self.n, _lib.ptr(z), cs, ps)
                else:
                    sl = None
                    try:
                        sl = self._slice(ev)
                    except Exception as e:                                       # noqa: BLE001 -- a wrong-sized slice on ONE rank: its peers learn it from the next collective
                        if not exchanging:
                            raise
                        state["failed"] = state["failed"] or e
                    guard(lib.kzg_commit_and_prove_lagrange_begin, self.ctx.handle, self.srs.handle, self.lo, _lib.ptr(sl) if self.len and sl is not None else None,
                          self.len, self.n, _lib.ptr(z), cs, ps)
                if pending_y:                                                    # (lag_y) the previous blob's y: fold, enqueue its quotient + MSM
                    finish_y()
                    if state["bad"]:
                        inflight.append((cs, ps, z, None))
                        drain_and_raise()
                # (buffer lifetime: the host slice `sl` is read by an asynchronous copy on the slot's stream; partial_y below waits for phase 1, which is
                # behind that copy, BEFORE the generator is asked for its next item -- a producer that refills one pinned buffer per blob is safe)
                ypart = np.zeros(16, dtype=np.uint64)                           # 8 words used: S_g | f_m
                guard(lib.kzg_compute_proof_lagrange_partial_y, self.ctx.handle, ps, _lib.ptr(ypart))
                y = np.zeros(4, dtype=np.uint64)
                inflight.append((cs, ps, z, y))
                start_y(ps, z, y, ypart)
                if not lag_y:
                    finish_y()
                    if state["bad"]:                                             # every rank sees the same rows: all stop here, in step
                        drain_and_raise()
                t += 1
            if pending_y:
                finish_y()
                if state["bad"]:
                    drain_and_raise()
            while inflight or pending_result:
                if pending_result:
                    out = collect_result()
                    if state["bad"]:
                        drain_and_raise()
                    yield out
                if inflight:
                    finish_oldest()
        finally:
            for cs, ps, _z, _y in inflight:                                  # a consumer that stopped early: nothing may stay in flight
                release(cs, ps)
            inflight.clear()
            settle()


class MultiKzg:
    """Several GPUs behind one handle in ONE process (`kzg_multi_*`): device g holds the SRS powers [g N / G, (g+1) N / G) and one
    host thread of the library drives it; partial sums are folded on the host.  No torch, no collective: what a Rust host binds
    to get the 8-GPU split of SURVEY.md 8e.  `device_ids` may repeat an id (several contexts on one GPU)."""

    def __init__(self, device_ids):
        ids = (C.c_int32 * len(device_ids))(*device_ids)
        h = C.c_void_p()
        rc = _lib.load().kzg_multi_create(ids, len(device_ids), C.byref(h))
        if rc != _lib.OK:
            raise ValueError(_lib.status_message(rc))
        self.handle = h

    def _check(self, rc):
        if rc != _lib.OK:
            raise ValueError(_lib.status_message(rc))

    def srs_upload(self, g1_points):
        pts = _lib.as_u64(g1_points, 8).reshape(-1, 8)
        self._check(_lib.load().kzg_multi_srs_upload(self.handle, _lib.ptr(pts), pts.shape[0]))

    def srs_generate(self, tau: int, n: int):
        from .fr import fr_from_int
        self._check(_lib.load().kzg_multi_srs_generate(self.handle, _lib.ptr(fr_from_int(tau)), n))

    def __len__(self):
        return _lib.load().kzg_multi_srs_len(self.handle)

    def cache_lagrange(self, n: int):
        """Shard the Lagrange basis of the first n powers by evaluation index over the devices (`kzg_multi_cache_lagrange`): from then on
        `commit_eval_form` / `compute_proof` of exactly n evaluations touch each device's own slice only."""
        self._check(_lib.load().kzg_multi_cache_lagrange(self.handle, n))

    def commit_coeff_form(self, coeffs):
        c = _lib.as_u64(coeffs, 4).reshape(-1, 4)
        out = np.zeros(8, dtype=np.uint64); inf = C.c_uint8(0)
        self._check(_lib.load().kzg_multi_commit_coeff_form(self.handle, _lib.ptr(c), len(c), _lib.ptr(out), C.byref(inf)))
        return out

    def commit_eval_form(self, evals):
        e = _lib.as_u64(evals, 4).reshape(-1, 4)
        out = np.zeros(8, dtype=np.uint64); inf = C.c_uint8(0)
        self._check(_lib.load().kzg_multi_commit_eval_form(self.handle, _lib.ptr(e), len(e), _lib.ptr(out), C.byref(inf)))
        return out

    def compute_proof(self, evals, z_fr, n_roots=None):
        e = _lib.as_u64(evals, 4).reshape(-1, 4)
        z = np.ascontiguousarray(_lib.as_u64(z_fr, 0).reshape(4))
        out = np.zeros(8, dtype=np.uint64); inf = C.c_uint8(0); y = np.zeros(4, dtype=np.uint64)
        self._check(_lib.load().kzg_multi_compute_proof(self.handle, _lib.ptr(e), len(e), len(e) if n_roots is None else n_roots, _lib.ptr(z),
                                                        _lib.ptr(out), C.byref(inf), _lib.ptr(y)))
        return out, y

    def scalars_upload(self, buffer_id: int, coeffs):
        """Keep polynomial `buffer_id` (0 .. 15) resident: device g stores its slice of the coefficients."""
        c = _lib.as_u64(coeffs, 4).reshape(-1, 4)
        self._check(_lib.load().kzg_multi_scalars_upload(self.handle, buffer_id, _lib.ptr(c), len(c)))

    def commit_resident_stream(self, buffer_ids):
        """Commitments of the uploaded polynomials buffer_ids[k], every device pipelining its own MSMs: (count, 8) affine points."""
        ids = (C.c_int32 * max(1, len(buffer_ids)))(*buffer_ids)
        out = np.zeros((len(buffer_ids), 8), dtype=np.uint64)
        inf = np.zeros(max(1, len(buffer_ids)), dtype=np.uint8)
        self._check(_lib.load().kzg_multi_commit_resident_stream(self.handle, ids, len(buffer_ids), _lib.ptr(out), inf.ctypes.data_as(_lib.u8p)))
        return out

    def close(self):
        if getattr(self, "handle", None):
            _lib.load().kzg_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
