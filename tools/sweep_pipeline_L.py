"""Pipelined (two-slot) 2^20 MSM throughput against the accumulate segment length L."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
n = 1 << 20
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, n, ctx=ctx)
d = torch.from_numpy(bench.blob_like_scalars(n, 123).view(np.int64)).cuda(); torch.cuda.synchronize()
out = np.zeros(16, np.uint64)
def pipe(steps):
    prev = None
    for i in range(steps):
        assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, i & 1) == 0
        if prev is not None:
            assert lib.kzg_msm_g1_srs_end(ctx.handle, prev, None, None, _lib.ptr(out)) == 0
        prev = i & 1
    assert lib.kzg_msm_g1_srs_end(ctx.handle, prev, None, None, _lib.ptr(out)) == 0
for L in [int(x) for x in os.environ.get("SWEEP_L", "64,72,80,85,90,96,104,112,120,128").split(",")]:
    ctx.set_msm_window(0, L)
    pipe(6)
    t0 = time.perf_counter(); pipe(40); dt = time.perf_counter() - t0
    print(f"L={L:4d}: {dt/40*1e3:.3f} ms/MSM pipelined", flush=True)
