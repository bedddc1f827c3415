"""Same-box experiments on k_msm_accumulate: resident wave slots (KZG_ACC_SLOTS), library variants (KZG_LIB_PATH).
Each configuration runs in a child process; prints the library's own HIP-event phase timings (one MSM at a time)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes as C, hashlib, os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
n = 1 << int(os.environ.get("EXP_LOG_N", "20"))
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") %% bench.FR
srs = k.SRS.generate(tau, n, ctx=ctx)
d = torch.from_numpy(bench.blob_like_scalars(n, 12345).view(np.int64)).cuda(); torch.cuda.synchronize()
out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
n = int(os.environ.get("EXP_PAIRS", n))          # MSM over the first EXP_PAIRS scalars of the SRS
for L in [int(x) for x in os.environ.get("EXP_SEG", "0").split(",")]:
    ctx.set_msm_window(0, L)
    for it in range(3):
        lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(out), C.byref(inf))
    lib.kzg_ctx_set_profiling(ctx.handle, 1)
    t0 = time.perf_counter(); reps = 8
    for it in range(reps):
        lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(out), C.byref(inf))
    wall = (time.perf_counter() - t0) / reps * 1e3
    ph = (C.c_double * 8)(); la = C.c_uint64(0); pa = C.c_uint64(0)
    lib.kzg_ctx_get_msm_profile(ctx.handle, ph, C.byref(la), C.byref(pa))
    lib.kzg_ctx_set_profiling(ctx.handle, 0)
    print("%%-28s seg=%%3d wall=%%.3f digits=%%.3f sort=%%.3f scat=%%.3f acc=%%.3f red1=%%.3f red2=%%.3f total=%%.3f" %% (os.environ.get("EXP_TAG", ""), L, wall, ph[0] / la.value, ph[1] / la.value, ph[2] / la.value, ph[4] / la.value, ph[5] / la.value, ph[6] / la.value, ph[7] / la.value), flush=True)
''' % ROOT
configs = []
for spec in sys.argv[1:]:
    env = dict(os.environ)
    tag = []
    for kv in spec.split(","):
        if not kv:
            continue
        k_, v = kv.split("=", 1)
        env[k_] = v
        tag.append(kv.replace("KZG_", ""))
    env["EXP_TAG"] = " ".join(tag)[:28]
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
