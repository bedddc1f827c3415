"""Accumulate-kernel balance sweep: segment length (number of waves) x LDS-forced cap on resident blocks per CU.
Each configuration runs in a child process (the LDS request is read once per process)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes as C, hashlib, os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
n = 1 << 20
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") %% bench.FR
srs = k.SRS.generate(tau, n, ctx=ctx)
d = torch.from_numpy(bench.blob_like_scalars(n, 12345).view(np.int64)).cuda(); torch.cuda.synchronize()
out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
for L in [int(x) for x in os.environ["SWEEP_L"].split(",")]:
    ctx.set_msm_window(0, L)
    for it in range(2):
        lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(out), C.byref(inf))
    lib.kzg_ctx_set_profiling(ctx.handle, 1)
    t0 = time.perf_counter(); reps = 6
    for it in range(reps):
        lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(out), C.byref(inf))
    wall = (time.perf_counter() - t0) / reps * 1e3
    ph = (C.c_double * 8)(); la = C.c_uint64(0); pa = C.c_uint64(0)
    lib.kzg_ctx_get_msm_profile(ctx.handle, ph, C.byref(la), C.byref(pa))
    lib.kzg_ctx_set_profiling(ctx.handle, 0)
    print("lds=%%s L=%%3d waves=%%5d wall=%%.3f acc=%%.3f fin=%%.3f total=%%.3f" %% (os.environ.get("KZG_ACC_LDS", "0"), L, (16 * n) // (64 * L), wall, ph[4] / la.value, ph[5] / la.value, ph[7] / la.value), flush=True)
''' % ROOT
for lds in os.environ.get("SWEEP_LDS", "0,52000,80000").split(","):
    env = dict(os.environ, KZG_ACC_LDS=lds, SWEEP_L=os.environ.get("SWEEP_L", "64,85,96,128,171,256"))
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
