"""Does the number of HIP streams created before the slot streams change the depth-4 throughput?  (HW queue mapping probe)"""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
from rust_kzg_bn254_amd.sharding import ShardedMsm
lib = _lib.load()
ndummy = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx = k.Context(0)
dummies = [k.Context(0) for _ in range(ndummy)]
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
n = 1 << 17
srs = k.SRS.generate(tau, n, ctx=ctx)
d = torch.from_numpy(bench.blob_like_scalars(n, 123).view(np.int64)).cuda(); torch.cuda.synchronize()
sh = ShardedMsm(ctx, n)
for depth in (2, 3, 4):
    list(sh.commit_stream(srs, [d.data_ptr()] * 8, depth=depth))
    t0 = time.perf_counter(); list(sh.commit_stream(srs, [d.data_ptr()] * 80, depth=depth)); dt = time.perf_counter() - t0
    print(f"dummy contexts {ndummy}: depth {depth}: {dt/80*1e3:.3f} ms/MSM", flush=True)
