import ctypes as C, hashlib, os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
n = 1 << 20
lib = _lib.load()
ctxs = [k.Context(0) for _ in range(3)]
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, n, ctx=ctxs[0])
sc = bench.blob_like_scalars(n, 123)
d = torch.from_numpy(sc.view(np.int64)).cuda(); torch.cuda.synchronize()
outs = [np.zeros(8, np.uint64) for _ in ctxs]
def one(i):
    inf = C.c_uint8(0)
    rc = lib.kzg_msm_g1_srs_device(ctxs[i].handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(outs[i]), C.byref(inf))
    assert rc == 0, rc
for inflight in (1, 2, 3):
    with ThreadPoolExecutor(inflight) as ex:
        list(ex.map(one, [j % inflight for j in range(6)]))
        steps = 30
        t0 = time.perf_counter()
        list(ex.map(one, [j % inflight for j in range(steps)]))
        dt = time.perf_counter() - t0
    print(f"in flight {inflight}: {dt/steps*1e3:.3f} ms/commit -> {steps/dt:.1f} commitments/s", flush=True)
assert all(np.array_equal(outs[0], o) for o in outs)
