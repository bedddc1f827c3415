"""Where does the one-time stall of the pipelined stream sit: after how many MSMs / how much time since the context started?"""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd.sharding import ShardedMsm
ctx = k.Context(0)
n = 1 << 20
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, n, ctx=ctx)
d = torch.from_numpy(bench.blob_like_scalars(n, 123).view(np.int64)).cuda(); torch.cuda.synchronize()
sh = ShardedMsm(ctx, n)
T0 = time.perf_counter()
count = 0
for length in [int(x) for x in os.environ.get("LENS", "2,5,5,5,5,20,20").split(",")]:
    ts = []
    t0 = time.perf_counter()
    for _ in sh.commit_stream(srs, [d.data_ptr()] * length, depth=2):
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    dt = [ts[0]] + [b - a for a, b in zip(ts, ts[1:])]
    print("stream of %2d (MSMs before: %3d, %.1f ms since start): %s" % (length, count, (t0 - T0) * 1e3, " ".join("%.2f" % (x * 1e3) for x in dt)), flush=True)
    count += length
    time.sleep(float(os.environ.get("GAP", "0")))
