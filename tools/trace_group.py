"""Diagnostic: completion times of the launches of a grouped shard stream (first-launch stall after a device-wide synchronisation)."""
import ctypes as C, hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd.sharding import ShardedMsm
ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
n = 1 << 17
srs = k.SRS.generate(tau, n, ctx=ctx)
bufs = [torch.from_numpy(bench.blob_like_scalars(n, 100 + j).view(np.int64)).cuda() for j in range(8)]
torch.cuda.synchronize()
ptrs = [bufs[j % 8].data_ptr() for j in range(48)]
sh = ShardedMsm(ctx, n)
def run(tag, depth, group, pre=None):
    if pre: pre()
    marks = []; t = time.perf_counter()
    for r in sh.commit_stream(srs, ptrs, depth=depth, group=group):
        t2 = time.perf_counter(); marks.append((t2 - t) * 1e3); t = t2
    print("%-34s" % tag, " ".join("%.2f" % m for m in marks if m > 0.005)[:110], flush=True)
run("alloc", 3, 4)
run("back to back d3 g4", 3, 4)
run("sync d3 g4", 3, 4, torch.cuda.synchronize)
run("sync d2 g4", 2, 4, torch.cuda.synchronize)
run("sync d3 g2", 3, 2, torch.cuda.synchronize)
run("sync d1 g4", 1, 4, torch.cuda.synchronize)
run("sleep 20ms d3 g4", 3, 4, lambda: time.sleep(0.02))
run("sync+sleep 20ms d3 g4", 3, 4, lambda: (torch.cuda.synchronize(), time.sleep(0.02)))
run("sync d3 g4 again", 3, 4, torch.cuda.synchronize)
def hipsync():
    assert k._lib.load().kzg_ctx_synchronize(ctx.handle) == 0 if hasattr(k._lib.load(), "kzg_ctx_synchronize") else True
run("back to back d3 g4", 3, 4)
