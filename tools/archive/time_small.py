"""Latency of small MSMs: generic mode (g1_lincomb, caller bases, e.g. batch verification) and table mode (SRS)."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
rng = np.random.default_rng(5)
names = ["digits|hist1", "sort1", "sort2", "-", "accumulate", "bucket_sums+bits1", "bits2|reduce", "device_total"]
for log_n in [int(x) for x in os.environ.get("SMALL_LOGS", "10,12,14,16,18").split(",")]:
    n = 1 << log_n
    uni = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); uni[:, 3] &= np.uint64((1 << 60) - 1)
    srs = k.SRS.generate(tau, n, ctx=ctx)
    pts = srs.g1
    out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
    for label, fn in (("table  ", lambda: lib.kzg_msm_g1_srs(ctx.handle, srs.handle, 0, _lib.ptr(uni), n, _lib.ptr(out), C.byref(inf))),
                      ("generic", lambda: lib.kzg_msm_g1(ctx.handle, _lib.ptr(pts), n, _lib.ptr(uni), n, _lib.ptr(out), C.byref(inf)))):
        for _ in range(3): fn()
        lib.kzg_ctx_set_profiling(ctx.handle, 1)
        t0 = time.perf_counter()
        for _ in range(10): fn()
        wall = (time.perf_counter() - t0) / 10 * 1e3
        ph = (C.c_double * 8)(); la = C.c_uint64(0); pa = C.c_uint64(0)
        lib.kzg_ctx_get_msm_profile(ctx.handle, ph, C.byref(la), C.byref(pa))
        lib.kzg_ctx_set_profiling(ctx.handle, 0)
        print(f"n=2^{log_n} {label} wall={wall:7.3f} ms | " + " ".join(f"{nm}={ph[i]/10:.3f}" for i, nm in enumerate(names)), flush=True)
    srs.close()

# BASELINE config 5 shape: three 4096-point MSMs, separately and batched
n = 4096
uni = rng.integers(0, 1 << 62, size=(3 * n, 4), dtype=np.uint64); uni[:, 3] &= np.uint64((1 << 60) - 1)
srs = k.SRS.generate(tau, 3 * n, ctx=ctx); pts = srs.g1
out = np.zeros((3, 8), np.uint64); inf = np.zeros(3, np.uint8)
def sep():
    for b in range(3):
        lib.kzg_msm_g1(ctx.handle, _lib.ptr(pts[b * n:(b + 1) * n]), n, _lib.ptr(uni[b * n:(b + 1) * n]), n, _lib.ptr(out[b]), None)
def bat():
    lib.kzg_msm_g1_batch(ctx.handle, _lib.ptr(pts), _lib.ptr(uni), n, 3, _lib.ptr(out), inf.ctypes.data_as(_lib.u8p))
for name, fn in (("3 x kzg_msm_g1(4096)", sep), ("kzg_msm_g1_batch(4096 x 3)", bat)):
    for _ in range(3): fn()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    print(f"{name}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
