"""Batched commitments (kzg_commit_coeff_form_batch_device): microseconds per commitment against one call per polynomial."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load()
ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << 16, ctx=ctx)
for n, count in ((512, 1024), (1024, 1024), (2048, 1024), (4096, 1024), (2048, 64), (2048, 4096), (8192, 256), (16384, 256), (32768, 64), (65536, 32)):
    sc = bench.blob_like_scalars(n * count, 77)
    d = torch.from_numpy(sc.view(np.int64)).cuda(); torch.cuda.synchronize()
    out = np.zeros((count, 8), np.uint64)
    def batch():
        rc = lib.kzg_commit_coeff_form_batch_device(ctx.handle, srs.handle, C.c_void_p(d.data_ptr()), n, count, _lib.ptr(out), None)
        assert rc == 0, rc
    batch(); batch()
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps): batch()
    dt = (time.perf_counter() - t0) / reps
    o1 = np.zeros(8, np.uint64); inf = C.c_uint8(0)
    m = min(count, 64)
    def singles():
        for j in range(m):
            assert lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr() + j * n * 32), n, _lib.ptr(o1), C.byref(inf)) == 0
    singles()
    t0 = time.perf_counter(); singles(); ds = (time.perf_counter() - t0) / m
    assert np.array_equal(o1, out[m - 1])
    print(f"n = {n:6d} x {count:5d} polynomials: batch {dt*1e3:8.3f} ms = {dt/count*1e6:7.2f} us per commitment ({count/dt:9.0f} /s, {n*count/dt:.3e} pairs/s); "
          f"one call per polynomial {ds*1e6:7.1f} us", flush=True)
