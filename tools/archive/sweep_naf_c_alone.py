"""Lone-commitment latency (resident scalars, one at a time) by NAF bucket bits: run once per KZG_NAF_C value (read at library load)."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << 20, ctx=ctx)
out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
import gc; gc.disable()
row = []
for log_n in (15, 16, 17, 18, 19, 20):
    n = 1 << log_n
    d = torch.from_numpy(bench.blob_like_scalars(n, 123).view(np.int64)).cuda(); torch.cuda.synchronize()
    def one():
        assert lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(out), C.byref(inf)) == 0
    for _ in range(40): one()
    ts = []
    for _ in range(60):
        t0 = time.perf_counter(); one(); ts.append(time.perf_counter() - t0)
    ts.sort()
    row.append("2^%d %.3f" % (log_n, ts[len(ts) // 2] * 1e3))
print("KZG_NAF_C=%s | median ms: " % os.environ.get("KZG_NAF_C", "auto") + "  ".join(row), flush=True)
