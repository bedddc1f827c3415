"""Per-step completion times of the pipelined commitment stream right after a set-up + short warm-up (what bench.py --steps 20 --warmup 5 sees)."""
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
depth = int(os.environ.get("DEPTH", "2"))
list(sh.commit_stream(srs, [d.data_ptr()] * depth, depth=depth))
list(sh.commit_stream(srs, [d.data_ptr()] * 5, depth=depth))
torch.cuda.synchronize()
for rep in range(3):
    ts = []
    t0 = time.perf_counter()
    for _ in sh.commit_stream(srs, [d.data_ptr()] * 40, depth=depth):
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    dt = [ts[0]] + [b - a for a, b in zip(ts, ts[1:])]
    print("rep %d total %.3f ms/step | first steps (ms): %s | last: %s" % (rep, ts[-1] / 40 * 1e3, " ".join("%.2f" % (x * 1e3) for x in dt[:12]), " ".join("%.2f" % (x * 1e3) for x in dt[-5:])), flush=True)
    time.sleep(float(os.environ.get("GAP", "0")))
