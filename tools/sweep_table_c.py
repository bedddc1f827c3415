"""Window bits of the SRS tables vs MSM time, one MSM at a time and three in flight (shard-sized MSMs of the 8-GPU split)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes as C, hashlib, os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
from rust_kzg_bn254_amd.sharding import ShardedMsm
log_n = int(os.environ["SW_LOG_N"]); n = 1 << log_n
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") %% bench.FR
srs = k.SRS.generate(tau, n, ctx=ctx)
d = torch.from_numpy(bench.blob_like_scalars(n, 12345).view(np.int64)).cuda(); torch.cuda.synchronize()
sh = ShardedMsm(ctx, n)
res = []
for depth in [int(x) for x in os.environ.get("SW_DEPTHS", "1,3").split(",")]:
    list(sh.commit_stream(srs, [d.data_ptr()] * 8, depth=depth))
    t0 = time.perf_counter(); list(sh.commit_stream(srs, [d.data_ptr()] * 40, depth=depth)); res.append((time.perf_counter() - t0) / 40 * 1e3)
print("n=2^%%d c=%%s depths %%s: %%s ms" %% (log_n, os.environ.get("KZG_TABLE_C", "default"), os.environ.get("SW_DEPTHS", "1,3"), " ".join("%%.3f" %% r for r in res)), flush=True)
''' % ROOT
for log_n in [int(x) for x in os.environ.get("SW_LOGS", "17,18,19").split(",")]:
    for c in os.environ.get("SW_CS", "0,13,14,15,16").split(","):
        env = dict(os.environ, SW_LOG_N=str(log_n))
        if c != "0":
            env["KZG_TABLE_C"] = c
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
