import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
n = 1 << 20
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, n, ctx=ctx)
rng = np.random.default_rng(5)
uni = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); uni[:, 3] &= np.uint64((1 << 60) - 1)
d = torch.from_numpy(uni.view(np.int64)).cuda()
names = ["digits", "hist+scan", "scatter", "segments", "accumulate", "bucket_fin", "reduce", "device_total"]
out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
for c, L in ((0, 96), (16, 32), (16, 16), (14, 96), (14, 48), (12, 96)):
    ctx.set_msm_window(c, L)
    for _ in range(2): lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(out), C.byref(inf))
    lib.kzg_ctx_set_profiling(ctx.handle, 1)
    for _ in range(3): lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(out), C.byref(inf))
    ph = (C.c_double * 8)(); la = C.c_uint64(0); pa = C.c_uint64(0)
    lib.kzg_ctx_get_msm_profile(ctx.handle, ph, C.byref(la), C.byref(pa)); lib.kzg_ctx_set_profiling(ctx.handle, 0)
    W = (255 + (c or 16) - 1) // (c or 16)
    print(f"c={c:2d} ({'tables' if c == 0 else 'generic, points L2/MALL resident'}) L={L:3d} entries={W*n/1e6:.1f}M | " + " ".join(f"{nm}={ph[i]/3:.3f}" for i, nm in enumerate(names)), flush=True)
