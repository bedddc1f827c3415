"""Shard-size MSM latency vs. number of MSMs in flight (one context = one stream + workspace each).
Models what one rank of an N-GPU sharded commitment does: its slice of the SRS, n/N scalars resident in HBM."""
import ctypes as C, hashlib, os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load()
ctxs = [k.Context(0) for _ in range(3)]
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
for log_n in [int(x) for x in os.environ.get("SHARD_LOGS", "17,18,19,20").split(",")]:
    n = 1 << log_n
    srs = k.SRS.generate(tau, n, ctx=ctxs[0])
    sc = bench.blob_like_scalars(n, 123)
    d = torch.from_numpy(sc.view(np.int64)).cuda(); torch.cuda.synchronize()
    outs = [np.zeros(16, np.uint64) for _ in ctxs]
    def one(i):
        rc = lib.kzg_msm_g1_srs_partial_device(ctxs[i].handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(outs[i]))
        assert rc == 0, rc
    for inflight in (1, 2, 3):
        with ThreadPoolExecutor(inflight) as ex:
            list(ex.map(one, [j % inflight for j in range(6)]))
            steps = 60
            t0 = time.perf_counter()
            list(ex.map(one, [j % inflight for j in range(steps)]))
            dt = time.perf_counter() - t0
        print(f"n=2^{log_n} in flight {inflight}: {dt/steps*1e3:.3f} ms/MSM", flush=True)
    # the library's own two-slot pipeline (one host thread), as bench.py uses it
    ctx = ctxs[0]
    def pipe(steps):
        prev = None
        for i in range(steps):
            assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, i & 1) == 0
            if prev is not None:
                assert lib.kzg_msm_g1_srs_end(ctx.handle, prev, None, None, _lib.ptr(outs[0])) == 0
            prev = i & 1
        assert lib.kzg_msm_g1_srs_end(ctx.handle, prev, None, None, _lib.ptr(outs[0])) == 0
    pipe(6)
    t0 = time.perf_counter(); pipe(60); dt = time.perf_counter() - t0
    print(f"n=2^{log_n} begin/end pipeline: {dt/60*1e3:.3f} ms/MSM", flush=True)
    from rust_kzg_bn254_amd.sharding import ShardedMsm
    sh = ShardedMsm(ctx, n)
    for depth in (2, 3, 4):
        list(sh.commit_stream(srs, [d.data_ptr()] * 8, depth=depth))
        t0 = time.perf_counter(); list(sh.commit_stream(srs, [d.data_ptr()] * 60, depth=depth)); dt = time.perf_counter() - t0
        print(f"n=2^{log_n} commit_stream depth {depth}: {dt/60*1e3:.3f} ms/MSM", flush=True)
    srs.close()
