"""Per-call latency of small SRS MSMs, 300 calls per size: median, p99, max and the calls slower than 5 x the median (on the GPU boxes of this
round: one call in ~1 500 takes 1-40 ms at a random position -- host scheduling, not a kernel; medians are what the docs quote)."""
import ctypes as C, hashlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << 17, ctx=ctx)
out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
for log_n in (9, 10, 11, 12, 13):
    n = 1 << log_n
    d = torch.from_numpy(bench.blob_like_scalars(n, 123).view(np.int64)).cuda(); torch.cuda.synchronize()
    def one():
        assert lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(out), C.byref(inf)) == 0
    ts = []
    for i in range(300):
        t0 = time.perf_counter(); one(); ts.append((time.perf_counter() - t0) * 1e3)
    s = sorted(ts)
    big = [(i, round(t, 2)) for i, t in enumerate(ts) if t > 5 * s[len(s) // 2]]
    print("2^%d median %.3f p99 %.3f max %.3f outliers %s" % (log_n, s[150], s[296], s[-1], big[:8]), flush=True)
