"""Off-domain compute_proof from host buffers, 2^9 .. 2^16 evaluations on a 2^17-point SRS: median ms per call (KZG_POLY_SMALL_LOG: the size the one-workgroup inversion kernel takes the chain to)."""
import ctypes as C, hashlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << 17, ctx=ctx)
o8 = np.zeros(8, np.uint64); o4 = np.zeros(4, np.uint64); oi = C.c_uint8(0)
row = []
for log_n in (9, 10, 11, 12, 13, 14, 16):
    n = 1 << log_n
    sc = bench.blob_like_scalars(n, 5); zq = np.ascontiguousarray(bench.blob_like_scalars(4, 99)[1])
    def f(): assert lib.kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(sc), n, None, n, _lib.ptr(zq), _lib.ptr(o8), C.byref(oi), _lib.ptr(o4)) == 0
    for _ in range(5): f()
    ts = []
    for _ in range(40):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    row.append("2^%d %.3f" % (log_n, sorted(ts)[20] * 1e3))
print("KZG_POLY_SMALL_LOG=%s off-domain proof ms: " % os.environ.get("KZG_POLY_SMALL_LOG", "default") + "  ".join(row), flush=True)
