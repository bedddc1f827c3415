"""A SHORT timed region of shard-sized MSM steps (what the driver's `bench.py --gpus 8 --steps 20` times per rank): total wall time of 20 resident
2^LOG_SLICE-pair steps for several launch schedules (steps per launch, launches in flight), no exchange.  Fits total = a + b * steps from the 20- and
96-step regions.  Usage (GPU box): python tools/probe_shard_region.py"""
import collections
import ctypes as C
import hashlib
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib

log_slice = int(os.environ.get("LOG_SLICE", "17"))
per = 1 << log_slice
lib = _lib.load()
ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, per, ctx=ctx)
sets = [torch.from_numpy(bench.blob_like_scalars(per, 5 + j).view(np.int64)).cuda() for j in range(4)]
torch.cuda.synchronize()
parts = np.zeros((8, 16), dtype=np.uint64)


def region(schedule, depth, trace=None):
    inflight = collections.deque()
    step = 0
    t0 = time.perf_counter()
    for kk, cnt in enumerate(schedule):
        if len(inflight) == depth:
            slot, c = inflight.popleft()
            rc = lib.kzg_msm_g1_srs_end_batch(ctx.handle, slot, c, None, None, _lib.ptr(parts)) if c > 1 else lib.kzg_msm_g1_srs_end(ctx.handle, slot, None, None, _lib.ptr(parts))
            assert rc == 0
            if trace is not None:
                trace.append(("end", time.perf_counter() - t0))
        slot = kk % depth
        if cnt > 1:
            arr = (C.c_void_p * cnt)(*[C.c_void_p(sets[(step + j) % 4].data_ptr()) for j in range(cnt)])
            rc = lib.kzg_msm_g1_srs_device_begin_batch(ctx.handle, srs.handle, 0, arr, per, cnt, slot)
        else:
            rc = lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, 0, C.c_void_p(sets[step % 4].data_ptr()), per, slot)
        assert rc == 0, rc
        if trace is not None:
            trace.append(("begin %d" % cnt, time.perf_counter() - t0))
        step += cnt
        inflight.append((slot, cnt))
    while inflight:
        slot, c = inflight.popleft()
        rc = lib.kzg_msm_g1_srs_end_batch(ctx.handle, slot, c, None, None, _lib.ptr(parts)) if c > 1 else lib.kzg_msm_g1_srs_end(ctx.handle, slot, None, None, _lib.ptr(parts))
        assert rc == 0
        if trace is not None:
            trace.append(("end", time.perf_counter() - t0))
    return time.perf_counter() - t0


def measure(name, schedule, depth):
    steps = sum(schedule)
    region(schedule, depth)
    ts = []
    for _ in range(15):
        torch.cuda.synchronize()
        ts.append(region(schedule, depth))
    med = statistics.median(ts)
    print("%-44s depth %d: %2d steps %.3f ms = %.4f ms per step (min %.3f)" % (name, depth, steps, med * 1e3, med / steps * 1e3, min(ts) * 1e3), flush=True)
    return med


cap = int(lib.kzg_msm_batch_capacity(per))
print("slice 2^%d, batch capacity %d" % (log_slice, cap))
g = min(4, cap)
t20 = measure("%d x %d" % (20 // g, g), [g] * (20 // g), 2)
t96 = measure("%d x %d" % (96 // g, g), [g] * (96 // g), 2)
b = (t96 - t20) / 76
print("   fit: %.3f ms + %.4f ms per step" % ((t20 - 20 * b) * 1e3, b * 1e3))
measure("%d x %d" % (20 // g, g), [g] * (20 // g), 3)
if g == 4:
    measure("2 4 4 4 4 2 (staggered)", [2, 4, 4, 4, 4, 2], 2)
    measure("1 3 4 4 4 4", [1, 3, 4, 4, 4, 4], 2)
    measure("2 2 4 4 4 4", [2, 2, 4, 4, 4, 4], 2)
    measure("4 4 4 4 2 2", [4, 4, 4, 4, 2, 2], 2)
    measure("2 2 4 4 4 2 2", [2, 2, 4, 4, 4, 2, 2], 2)
    measure("2 x 10", [2] * 10, 2)
    measure("2 x 10", [2] * 10, 3)
    measure("2 x 10", [2] * 10, 4)
    measure("3 3 3 3 3 3 2", [3, 3, 3, 3, 3, 3, 2], 3)
    measure("1 x 20", [1] * 20, 3)
    measure("1 x 20", [1] * 20, 4)
tr = []
region([g] * (20 // g), 2, tr)
print("trace of one %d x %d region (ms):" % (20 // g, g), " ".join("%s@%.3f" % (a, t * 1e3) for a, t in tr))
