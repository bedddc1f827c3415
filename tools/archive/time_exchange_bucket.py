"""Cost of the per-step RCCL exchange in ShardedMsm.commit_stream on one rank (world-1 communicator, the collective forced):
shard-sized MSMs, three in flight, for several bucket sizes (steps per all-gather) against no exchange at all."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
import numpy as np, torch, torch.distributed as dist
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import sharding
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
class Forced(sharding.ShardedMsm):           # world > 1 code path with a one-rank communicator
    pass
for log_n in (17, 18, 19):
    n = 1 << log_n
    srs = k.SRS.generate(tau, n, ctx=ctx)
    d = torch.from_numpy(bench.blob_like_scalars(n, 9).view(np.int64)).cuda(); torch.cuda.synchronize()
    plain = sharding.ShardedMsm(ctx, n)
    forced = Forced(ctx, n, 0, 1)
    forced.world = 2                          # take the exchange branch ...
    forced._gatherer = sharding.PartialGatherer(1, "cuda")     # ... over the one-rank communicator
    def run(sh, steps, **kw):
        list(sh.commit_stream(srs, [d.data_ptr()] * 8, **kw))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = list(sh.commit_stream(srs, [d.data_ptr()] * steps, **kw))
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, outs[-1]
    base, want = run(plain, 120, depth=3)
    row = ["no exchange %.3f" % base]
    for b in (1, 2, 4, 8):
        ms, got = run(forced, 120, depth=3, bucket=b)
        assert np.array_equal(got, want)
        row.append("bucket %d: %.3f" % (b, ms))
    print("2^%d pairs, depth 3, ms per step: " % log_n + "  ".join(row), flush=True)
    srs.close()
dist.destroy_process_group()
