"""Diagnostic: where the first launch of a grouped shard stream loses 6-7 ms (host time inside begin / end calls per launch)."""
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

class Timed(ShardedMsm):
    log = []
    def begin_group(self, s, p, slot):
        t = time.perf_counter(); r = super().begin_group(s, p, slot); self.log.append(("begin%d" % len(p), (time.perf_counter() - t) * 1e3)); return r
    def begin(self, s, p, slot):
        t = time.perf_counter(); r = super().begin(s, p, slot); self.log.append(("begin1", (time.perf_counter() - t) * 1e3)); return r
    def _end_group(self, slot, count):
        t = time.perf_counter(); r = super()._end_group(slot, count); self.log.append(("end%d" % count, (time.perf_counter() - t) * 1e3)); return r

sh = Timed(ctx, n)
def run(tag, steps, depth, group, pre=None):
    if pre: pre()
    sh.log.clear()
    t = time.perf_counter()
    list(sh.commit_stream(srs, ptrs[:steps], depth=depth, group=group))
    print("%-26s total %.2f ms: %s" % (tag, (time.perf_counter() - t) * 1e3, " ".join("%s=%.2f" % e for e in sh.log[:8])), flush=True)
run("alloc", 48, 2, 4)
run("warm 5", 5, 2, 4)
run("sync, 20 steps", 20, 2, 4, torch.cuda.synchronize)
run("warm 8", 8, 2, 4)
run("sync, 20 steps", 20, 2, 4, torch.cuda.synchronize)
run("warm 5", 5, 2, 4)
run("no sync, 20 steps", 20, 2, 4)
run("warm 5", 5, 2, 4)
run("sync, 20 steps, g1 d3", 20, 3, 1, torch.cuda.synchronize)
