"""How much do more MSMs in flight help at shard sizes?  One context (stream + workspace) per in-flight MSM, one host thread each."""
import ctypes as C, hashlib, os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load()
NC = 6
ctxs = [k.Context(0) for _ in range(NC)]
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
for log_n in (17, 18):
    n = 1 << log_n
    srs = k.SRS.generate(tau, n, ctx=ctxs[0])
    d = torch.from_numpy(bench.blob_like_scalars(n, 123).view(np.int64)).cuda(); torch.cuda.synchronize()
    for inflight in (1, 2, 3, 4, 6):
        steps_per = 40
        def worker(i):
            out = np.zeros(16, np.uint64)
            for _ in range(steps_per):
                rc = lib.kzg_msm_g1_srs_partial_device(ctxs[i].handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(out))
                assert rc == 0
        ths = [threading.Thread(target=worker, args=(i,)) for i in range(inflight)]
        for i in range(inflight): worker_warm = lib.kzg_msm_g1_srs_partial_device(ctxs[i].handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(np.zeros(16, np.uint64)))
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.perf_counter() - t0
        print(f"n=2^{log_n} threads/contexts in flight {inflight}: {dt/(steps_per*inflight)*1e3:.3f} ms/MSM", flush=True)
    srs.close()
