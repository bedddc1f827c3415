"""Shard-size commitments as a stream on one rank: one launch per step against several steps per launch (ShardedMsm.commit_stream group=)."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd.sharding import ShardedMsm
ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
for log_n in [int(x) for x in os.environ.get("SHARD_LOGS", "15,16,17,18").split(",")]:
    n = 1 << log_n
    srs = k.SRS.generate(tau, n, ctx=ctx)
    bufs = [torch.from_numpy(bench.blob_like_scalars(n, 100 + j).view(np.int64)).cuda() for j in range(8)]
    torch.cuda.synchronize()
    ptrs = [bufs[j % 8].data_ptr() for j in range(96)]
    sh = ShardedMsm(ctx, n)
    ref = None
    for depth, group in ((3, 1), (2, 2), (3, 2), (2, 4), (3, 4), (2, 8)):
        try:
            outs = list(sh.commit_stream(srs, ptrs[:16], depth=depth, group=group))
            t0 = time.perf_counter(); outs = list(sh.commit_stream(srs, ptrs, depth=depth, group=group)); dt = time.perf_counter() - t0
        except ValueError as e:
            print(f"n=2^{log_n} depth {depth} group {group}: {e}", flush=True)
            continue
        if ref is None:
            ref = outs
        same = all(np.array_equal(a, b) for a, b in zip(outs, ref))
        print(f"n=2^{log_n} depth {depth} group {group}: {dt/len(ptrs)*1e3:.4f} ms per step, same results {same}", flush=True)
    srs.close()
