"""Shard-sized MSMs with 4 in flight: ms per MSM against the segment length L (one lane accumulates L entries)."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
from rust_kzg_bn254_amd.sharding import ShardedMsm
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
for log_n in (17, 18, 19):
    n = 1 << log_n
    srs = k.SRS.generate(tau, n, ctx=ctx)
    d = torch.from_numpy(bench.blob_like_scalars(n, 123).view(np.int64)).cuda(); torch.cuda.synchronize()
    sh = ShardedMsm(ctx, n)
    for L in [int(x) for x in os.environ.get("SWEEP_L", "0,16,24,32,48,64,96").split(",")]:
        ctx.set_msm_window(0, L)
        for depth in (1, 4):
            list(sh.commit_stream(srs, [d.data_ptr()] * 8, depth=depth))
            t0 = time.perf_counter(); list(sh.commit_stream(srs, [d.data_ptr()] * 60, depth=depth)); dt = time.perf_counter() - t0
            print(f"n=2^{log_n} L={L:3d} depth {depth}: {dt/60*1e3:.3f} ms/MSM", flush=True)
    ctx.set_msm_window(0, 0)
    srs.close()
