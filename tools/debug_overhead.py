import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
from rust_kzg_bn254_amd.sharding import ShardedMsm
n = 1 << 20
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, n, ctx=ctx)
sc = bench.blob_like_scalars(n, 0x4B5A472D424E3235 & 0x7FFFFFFF)
d = torch.from_numpy(sc.view(np.int64)).cuda(); torch.cuda.synchronize()
sh = ShardedMsm(ctx, n, 0, 1)
out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
def direct(): lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(out), C.byref(inf))
def via_sh(): sh.commit_device(srs, d.data_ptr())
for name, fn, prof in (("direct", direct, 0), ("direct+prof", direct, 1), ("sharded", via_sh, 0), ("sharded+prof", via_sh, 1), ("direct", direct, 0)):
    lib.kzg_ctx_set_profiling(ctx.handle, prof)
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{name:14s} {dt:.3f} ms/step", flush=True)
