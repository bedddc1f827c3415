"""Where a rank's time goes in the config-4 stream (sharding.commit_and_prove_stream) at one slice size: the same schedule of C-ABI calls, no
exchange (one rank's rows folded as they are), host wall time per call class over 32 blobs.  LOG_SLICE (default 17), DEPTH (default 2).
Usage (GPU box): python tools/trace_config4_stream.py"""
import collections
import ctypes as C
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib

LOG_N = 20
n = 1 << LOG_N
per = 1 << int(os.environ.get("LOG_SLICE", "17"))
depth = int(os.environ.get("DEPTH", "2"))
grouped = os.environ.get("GROUPED", "0") == "1"          # one slot per blob, commitment + proof as one batched launch
lib = _lib.load()
ctx = k.Context(0)
P = _lib.ptr
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
os.environ["KZG_NO_PRECOMPUTE"] = "1"
plain = k.SRS.generate(tau, n, ctx=ctx)
del os.environ["KZG_NO_PRECOMPUTE"]
lag = plain.lagrange_shard(n, 0, per)
plain.close()
sets = [torch.from_numpy(bench.blob_like_scalars(per, 5 + j).view(np.int64)).cuda() for j in range(4)]
torch.cuda.synchronize()
z = np.ascontiguousarray(bench.uniform_scalars(4, 99)[1][1])
T = collections.defaultdict(float)


def timed(name, fn, *a):
    t0 = time.perf_counter()
    rc = fn(*a)
    T[name] += time.perf_counter() - t0
    assert rc == 0, (name, rc)


def finish(ocs, ops):
    cpart = np.zeros(16, np.uint64); ppart = np.zeros(32, np.uint64)
    if grouped:
        timed("end commit + proof (wait)", lib.kzg_commit_and_prove_lagrange_end, ctx.handle, ops, P(cpart), P(ppart))
    else:
        timed("end commit (wait)", lib.kzg_msm_g1_srs_end, ctx.handle, ocs, None, None, P(cpart))
        timed("end proof (wait)", lib.kzg_compute_proof_lagrange_end, ctx.handle, ops, P(ppart))


def run(count):
    inflight = collections.deque()
    for t in range(count):
        cs, ps = ((t % depth, t % depth) if grouped else ((2 * t) % (2 * depth), (2 * t + 1) % (2 * depth)))
        if len(inflight) == depth:
            finish(*inflight.popleft())
        timed("begin (enqueue phase 1 + commit MSM)", lib.kzg_commit_and_prove_lagrange_begin_device, ctx.handle, lag.handle, 0, C.c_void_p(sets[t % 4].data_ptr()), per, n, P(z), cs, ps)
        yp = np.zeros(8, np.uint64); y = np.zeros(4, np.uint64)
        timed("partial_y (wait for phase 1)", lib.kzg_compute_proof_lagrange_partial_y, ctx.handle, ps, P(yp))
        timed("fold y (host)", lib.kzg_lagrange_fold_y, P(yp), 1, n, P(z), P(y))
        timed("continue (enqueue quotient + MSM)", lib.kzg_compute_proof_lagrange_continue, ctx.handle, ps, P(y))
        inflight.append((cs, ps))
    while inflight:
        finish(*inflight.popleft())


run(8)
T.clear()
torch.cuda.synchronize()
t0 = time.perf_counter()
N_BLOBS = 32
run(N_BLOBS)
torch.cuda.synchronize()
total = time.perf_counter() - t0
print("slice 2^%d, depth %d%s: %.3f ms per blob (commitment + proof), no exchange" % (per.bit_length() - 1, depth, ", grouped launches" if grouped else "", total / N_BLOBS * 1e3))
for name, v in T.items():
    print("   %-40s %.3f ms per blob" % (name, v / N_BLOBS * 1e3))
