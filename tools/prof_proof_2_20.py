"""The proof / blob pipelines at the BASELINE size under rocprofv3 --kernel-trace: WHAT selects ONE entry point so that
every run's kernel_stats.csv belongs to one pipeline --
  proof_off   kzg_compute_proof, z off the domain      (prover/src/kzg.rs:128-178)
  proof_on    kzg_compute_proof, z = w^m               (kzg.rs:237-260)
  proof_lag   kzg_compute_proof over the cached Lagrange basis (no INTT; kzg.rs:96-100 applied to the quotient)
  commit_eval kzg_commit_eval_form                     (kzg.rs:84-104)
  commit_blob kzg_commit_blob                          (kzg.rs:182-185)
  proof_stream / commit_stream   the two-slot streamed forms of bench.py's host_buffers_*_streamed_ms
Prints the wall time per call; the kernel table comes from rocprofv3 (tools/prof_proof_2_20.sh)."""
import ctypes as C
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (one HIP runtime: INTEGRATION.md section 5)
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib

what = os.environ.get("WHAT", "proof_off")
log_n = int(os.environ.get("LOG_N", "20"))
reps = int(os.environ.get("REPS", "10"))
n = 1 << log_n
lib = _lib.load()
ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, n, ctx=ctx)
sc = bench.blob_like_scalars(n, 5)
o8 = np.zeros(8, np.uint64); o4 = np.zeros(4, np.uint64); oi = C.c_uint8(0)
z_off = np.ascontiguousarray(bench.blob_like_scalars(4, 99)[1])
roots = np.zeros((n, 4), np.uint64); nr = C.c_size_t(0)
assert lib.kzg_calculate_roots_of_unity(ctx.handle, n * 32, _lib.ptr(roots), n, C.byref(nr)) == 0
z_on = np.ascontiguousarray(roots[(n * 3) // 7])
u8p = C.POINTER(C.c_uint8)
blob = np.frombuffer(b"".join(b"\x00" + bytes(r) for r in np.random.default_rng(7).integers(32, 127, size=(n, 31), dtype=np.uint8)), dtype=np.uint8).copy()


def proof(z):
    assert lib.kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(sc), n, None, n, _lib.ptr(z), _lib.ptr(o8), C.byref(oi), _lib.ptr(o4)) == 0


def stream(begin, end, count):
    prev = None
    for i in range(count):
        assert begin(i & 1) == 0
        if prev is not None:
            assert end(prev) == 0
        prev = i & 1
    assert end(prev) == 0


if what in ("proof_lag", "commit_eval_lag", "commit_blob_lag"):
    t = time.perf_counter()
    assert lib.kzg_srs_cache_lagrange(ctx.handle, srs.handle, n) == 0
    print("kzg_srs_cache_lagrange(2^%d) %.1f ms" % (log_n, (time.perf_counter() - t) * 1e3), flush=True)
fns = {
    "proof_off": lambda: proof(z_off),
    "proof_lag": lambda: proof(z_off),
    "proof_on": lambda: proof(z_on),
    "commit_eval_lag": lambda: lib.kzg_commit_eval_form(ctx.handle, srs.handle, _lib.ptr(sc), n, _lib.ptr(o8), C.byref(oi)),
    "commit_blob_lag": lambda: lib.kzg_commit_blob(ctx.handle, srs.handle, blob.ctypes.data_as(u8p), blob.size, _lib.ptr(o8), C.byref(oi)),
    "commit_eval": lambda: lib.kzg_commit_eval_form(ctx.handle, srs.handle, _lib.ptr(sc), n, _lib.ptr(o8), C.byref(oi)),
    "commit_blob": lambda: lib.kzg_commit_blob(ctx.handle, srs.handle, blob.ctypes.data_as(u8p), blob.size, _lib.ptr(o8), C.byref(oi)),
    "proof_stream": lambda: stream(lambda s: lib.kzg_compute_proof_begin(ctx.handle, srs.handle, _lib.ptr(sc), n, None, n, _lib.ptr(z_off), s),
                                   lambda s: lib.kzg_compute_proof_end(ctx.handle, s, _lib.ptr(o8), C.byref(oi), _lib.ptr(o4)), 8),
    "commit_stream": lambda: stream(lambda s: lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, 0, _lib.ptr(sc), n, s),
                                    lambda s: lib.kzg_msm_g1_srs_end(ctx.handle, s, _lib.ptr(o8), C.byref(oi), None), 8),
}
f = fns[what]
per = 8 if what.endswith("_stream") else 1
for _ in range(2):
    f()
t = time.perf_counter()
for _ in range(reps):
    f()
print("%s 2^%d: %.3f ms per call (%d calls)" % (what, log_n, (time.perf_counter() - t) / (reps * per) * 1e3, reps * per), flush=True)
