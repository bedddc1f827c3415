"""The reference's small bench shapes (512 .. 2 048 coefficients, prover/benches/bench_kzg_commit.rs, bench_kzg_proof.rs) one call at a time under
rocprofv3 --kernel-trace: which kernels a call launches and how much of its wall time they cover.  WHAT = commit | proof | proof_off; N (default 512).
Usage (GPU box): cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o run -- python3 $REPO/tools/trace_small_shapes.py"""
import ctypes as C
import hashlib
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib

what = os.environ.get("WHAT", "commit")
n = int(os.environ.get("N", "512"))
lib = _lib.load(); ctx = k.Context(0); P = _lib.ptr
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << 17, ctx=ctx)
sc = np.ascontiguousarray(bench.blob_like_scalars(n, 5))
o8 = np.zeros(8, np.uint64); o4 = np.zeros(4, np.uint64); oi = C.c_uint8(0)
roots = np.zeros((n, 4), np.uint64); nr = C.c_size_t(0)
assert lib.kzg_calculate_roots_of_unity(ctx.handle, n * 32, P(roots), n, C.byref(nr)) == 0
z_on = np.ascontiguousarray(roots[(n * 3) // 7]); z_off = np.ascontiguousarray(bench.uniform_scalars(4, 99)[1][1])
calls = {"commit": lambda: lib.kzg_commit_coeff_form(ctx.handle, srs.handle, P(sc), n, P(o8), C.byref(oi)),
         "proof": lambda: lib.kzg_compute_proof(ctx.handle, srs.handle, P(sc), n, None, n, P(z_on), P(o8), C.byref(oi), P(o4)),
         "proof_off": lambda: lib.kzg_compute_proof(ctx.handle, srs.handle, P(sc), n, None, n, P(z_off), P(o8), C.byref(oi), P(o4))}
f = calls[what]
for _ in range(10):
    assert f() == 0
ts = []
for _ in range(100):
    t0 = time.perf_counter(); rc = f(); ts.append(time.perf_counter() - t0)
    assert rc == 0
print("%s n = %d: median %.1f us, min %.1f us per call (110 calls in the trace)" % (what, n, statistics.median(ts) * 1e6, min(ts) * 1e6), flush=True)
