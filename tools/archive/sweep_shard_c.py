"""Shard-sized MSMs (2^17..2^19 pairs), two-slot pipeline: ms per MSM against the table window bits c (KZG_TABLE_C), child per c."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes as C, hashlib, os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") %% bench.FR
out = np.zeros(16, np.uint64)
for log_n in [int(x) for x in os.environ.get("SHARD_LOGS", "17,18,19").split(",")]:
    n = 1 << log_n
    srs = k.SRS.generate(tau, n, ctx=ctx)
    d = torch.from_numpy(bench.blob_like_scalars(n, 123).view(np.int64)).cuda(); torch.cuda.synchronize()
    def pipe(steps):
        prev = None
        for i in range(steps):
            assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, i & 1) == 0
            if prev is not None:
                assert lib.kzg_msm_g1_srs_end(ctx.handle, prev, None, None, _lib.ptr(out)) == 0
            prev = i & 1
        assert lib.kzg_msm_g1_srs_end(ctx.handle, prev, None, None, _lib.ptr(out)) == 0
    from rust_kzg_bn254_amd.sharding import ShardedMsm
    sh = ShardedMsm(ctx, n)
    res = []
    for depth in (3, 4):
        list(sh.commit_stream(srs, [d.data_ptr()] * 8, depth=depth))
        t0 = time.perf_counter(); list(sh.commit_stream(srs, [d.data_ptr()] * 80, depth=depth)); dt = time.perf_counter() - t0
        res.append(dt / 80 * 1e3)
    print("c=%%s n=2^%%d pipelined depth3 %%.3f depth4 %%.3f ms/MSM" %% (os.environ.get("KZG_TABLE_C", "auto"), log_n, res[0], res[1]), flush=True)
    srs.close()
''' % ROOT
for c in os.environ.get("SWEEP_C", "auto,11,12,13,14,15,16").split(","):
    env = dict(os.environ)
    if c != "auto":
        env["KZG_TABLE_C"] = c
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
