"""Per-rank time of BASELINE config 4 (2^20 evaluations in host memory -> commitment, proof) on ONE GPU at the slice sizes of 1 / 2 / 4 / 8 ranks,
round 4's replicated path against round 5's evaluation-index shards of the Lagrange basis ("a same-box table of
per-rank time at 2^17-element slices, old path vs new"):
  replicated  kzg_commit_eval_form_partial / kzg_compute_proof_partial: the rank uploads, transforms and divides the WHOLE polynomial, MSM over its slice
  lagrange    kzg_commit_eval_form_lagrange_partial / kzg_compute_proof_lagrange_{begin, partial_y, continue, end}: the rank's slice only
What rank 0 of a G-rank run executes per call, exchanges excluded (64 + 256 + 128 bytes per rank: latency only), one call at a time.
WHAT=rank8 (for rocprofv3): only the 2^17-slice calls of the Lagrange path, so that the kernel table belongs to one shape.
Usage (GPU box): python tools/time_config4_shards.py"""
import ctypes as C
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib

LOG_N = int(os.environ.get("LOG_N", "20"))
n = 1 << LOG_N
what = os.environ.get("WHAT", "table")
lib = _lib.load()
ctx = k.Context(0)
P = _lib.ptr
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
ev = bench.blob_like_scalars(n, 5)
z = np.ascontiguousarray(bench.uniform_scalars(4, 99)[1][1])
os.environ["KZG_NO_PRECOMPUTE"] = "1"
plain = k.SRS.generate(tau, n, ctx=ctx)
del os.environ["KZG_NO_PRECOMPUTE"]
full_lag = None


def med(fn, reps=12, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


rows = []
for G in ((8,) if what == "rank8" else (1, 2, 4, 8)):
    ln = n // G
    lag = plain.lagrange_shard(n, 0, ln)                        # rank 0's shard of the basis (g1_ifft once per call here: set-up, untimed)
    mono = k.SRS.generate(tau, ln, ctx=ctx)                     # rank 0's shard of the monomial SRS (the replicated path)
    sl = np.ascontiguousarray(ev[:ln])
    part = np.zeros(16, np.uint64); yp = np.zeros(8, np.uint64); pp = np.zeros(32, np.uint64); y = np.zeros(4, np.uint64)

    def new_commit():
        assert lib.kzg_commit_eval_form_lagrange_partial(ctx.handle, lag.handle, P(sl), ln, P(part)) == 0

    phases = [0.0, 0.0]

    def new_proof():
        t0 = time.perf_counter()
        assert lib.kzg_compute_proof_lagrange_begin(ctx.handle, lag.handle, 0, P(sl), ln, n, P(z), 0) == 0
        assert lib.kzg_compute_proof_lagrange_partial_y(ctx.handle, 0, P(yp)) == 0
        t1 = time.perf_counter()
        assert lib.kzg_lagrange_fold_y(P(yp), 1, n, P(z), P(y)) == 0          # (one rank's part: the value is not y, the work is the same)
        assert lib.kzg_compute_proof_lagrange_continue(ctx.handle, 0, P(y)) == 0
        assert lib.kzg_compute_proof_lagrange_end(ctx.handle, 0, P(pp)) == 0
        phases[0] += t1 - t0; phases[1] += time.perf_counter() - t1

    def old_commit():
        assert lib.kzg_commit_eval_form_partial(ctx.handle, mono.handle, 0, P(ev), n, P(part)) == 0

    def old_proof():
        assert lib.kzg_compute_proof_partial(ctx.handle, mono.handle, 0, P(ev), n, None, n, P(z), P(part), P(y)) == 0

    nc, np_ = med(new_commit), None
    phases[:] = [0.0, 0.0]
    np_ = med(new_proof, warm=0, reps=15)
    ph1, ph2 = phases[0] / 15 * 1e3, phases[1] / 15 * 1e3
    if what == "rank8":
        print("lagrange shard of 2^%d: commit %.3f ms, proof %.3f ms (phase 1 %.3f, phase 2 %.3f)" % (LOG_N - 3, nc, np_, ph1, ph2))
        break
    oc, op = med(old_commit), med(old_proof)
    rows.append((G, ln, oc, op, nc, np_, ph1, ph2))
    lag.close(); mono.close()

if rows:
    print("| ranks | slice | replicated (r4): commit / proof ms | Lagrange shards (r5): commit / proof ms (phase 1 + phase 2) | r4 / r5, commit + proof |")
    print("|---|---|---|---|---|")
    for G, ln, oc, op, nc, np_, ph1, ph2 in rows:
        print("| %d | 2^%d | %.3f / %.3f | %.3f / %.3f (%.3f + %.3f) | %.2f x |" % (G, ln.bit_length() - 1, oc, op, nc, np_, ph1, ph2, (oc + op) / (nc + np_)))
