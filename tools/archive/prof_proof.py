"""10 compute_proof calls (2^20 evaluations from host buffers, off-domain z) and 10 eval-form commitments, for rocprofv3 --kernel-trace."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
n = 1 << int(os.environ.get("LOG_N", "20"))
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, n, ctx=ctx)
sc = bench.blob_like_scalars(n, 5)
o8 = np.zeros(8, np.uint64); o4 = np.zeros(4, np.uint64); oi = C.c_uint8(0)
zq = np.ascontiguousarray(bench.blob_like_scalars(4, 99)[1])
def proof():
    assert lib.kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(sc), n, None, n, _lib.ptr(zq), _lib.ptr(o8), C.byref(oi), _lib.ptr(o4)) == 0
def ce():
    assert lib.kzg_commit_eval_form(ctx.handle, srs.handle, _lib.ptr(sc), n, _lib.ptr(o8), C.byref(oi)) == 0
for f, name in ((proof, "compute_proof"), (ce, "commit_eval_form")):
    for _ in range(2): f()
    t = time.perf_counter()
    for _ in range(10): f()
    print("%s %.3f ms" % (name, (time.perf_counter() - t) / 10 * 1e3), flush=True)
