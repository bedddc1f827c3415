"""Per-phase device times (the library's HIP events) and wall time of ONE commitment at a time, scalars resident, for small and
shard-sized MSMs over a loaded 2^20-point SRS: where does the latency of a small commitment go.  argv: log sizes (default 11 12 13 15 17)."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << int(os.environ.get("SRS_LOG", "20")), ctx=ctx)
out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
names = ["digits", "sort", "scatter", "-", "accumulate", "bits1", "bits2", "device"]
for log_n in [int(a) for a in sys.argv[1:]] or [11, 12, 13, 15, 17]:
    n = 1 << log_n
    d = torch.from_numpy(bench.blob_like_scalars(n, 123).view(np.int64)).cuda(); torch.cuda.synchronize()
    def one():
        assert lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(out), C.byref(inf)) == 0
    for _ in range(10): one()
    t0 = time.perf_counter()
    for _ in range(50): one()
    wall = (time.perf_counter() - t0) / 50 * 1e3
    lib.kzg_ctx_set_profiling(ctx.handle, 1)
    for _ in range(20): one()
    ph = (C.c_double * 8)(); la = C.c_uint64(0); pa = C.c_uint64(0)
    lib.kzg_ctx_get_msm_profile(ctx.handle, ph, C.byref(la), C.byref(pa))
    lib.kzg_ctx_set_profiling(ctx.handle, 0)
    print("2^%d wall %.3f ms | " % (log_n, wall) + "  ".join("%s %.1f" % (nm, ph[i] / la.value * 1e3) for i, nm in enumerate(names) if nm != "-") + " us", flush=True)
