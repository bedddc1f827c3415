import ctypes as C, hashlib, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << 20, ctx=ctx)
out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
for n in (1 << 18, (1 << 18) + 4096, (1 << 18) - 4096, 3 << 16, 1 << 17, 1 << 19):
    d = torch.from_numpy(bench.blob_like_scalars(n, 123).view(np.int64)).cuda(); torch.cuda.synchronize()
    ts = []
    for _ in range(24):
        t0 = time.perf_counter()
        assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, 0) == 0
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, _lib.ptr(out), C.byref(inf), None) == 0
        ts.append((time.perf_counter() - t0) * 1e3)
    print(n, " ".join("%.2f" % t for t in ts), flush=True)
    lib.kzg_ctx_set_profiling(ctx.handle, 1)
    for _ in range(8):
        assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, 0) == 0
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, _lib.ptr(out), C.byref(inf), None) == 0
    phase = (C.c_double * 8)(); launches, pairs = C.c_uint64(0), C.c_uint64(0)
    lib.kzg_ctx_get_msm_profile(ctx.handle, phase, C.byref(launches), C.byref(pairs))
    lib.kzg_ctx_set_profiling(ctx.handle, 0)
    print("   phases ms per launch:", ["%.3f" % (phase[i] / max(1, launches.value)) for i in range(8)], launches.value)
