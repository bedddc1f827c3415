"""The per-rank MSM stream of an N = 8 run (2^17-pair steps, four per launch, two launches in flight) through a ONE-RANK RCCL group: what the exchange
costs per step with the gatherer's copies and RCCL's own stream at normal / high priority (measured round 5: no difference beyond noise -- 0.167 / 0.167 /
0.164 / 0.166 ms per step over 96 steps, 0.158 without exchange; the ~0.1 ms of an exchange are host-side torch calls and the wait for the last one).
Usage (GPU box): python tools/probe_exchange_prio.py"""
import hashlib
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import sharding
from rust_kzg_bn254_amd.sharding import ShardedMsm

log_slice = int(os.environ.get("LOG_SLICE", "17"))
per = 1 << log_slice
ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, per, ctx=ctx)
sets = [torch.from_numpy(bench.blob_like_scalars(per, 5 + j).view(np.int64)).cuda() for j in range(4)]
ptrs = [t.data_ptr() for t in sets]
torch.cuda.synchronize()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")


def region(shm, count, depth, group):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in shm.commit_stream(srs, [ptrs[i % 4] for i in range(count)], depth=depth, group=group):
        pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / count * 1e3


def run(label, shm):
    grp = shm.auto_group(srs)
    dep = shm.group_depth(3, grp)
    region(shm, 48, dep, grp)
    t20 = sorted(region(shm, 20, dep, grp) for _ in range(9))[4]
    t96 = sorted(region(shm, 96, dep, grp) for _ in range(5))[2]
    print("%-58s group %d depth %d: 20 steps %.4f, 96 steps %.4f ms per step" % (label, grp, dep, t20, t96), flush=True)


run("no exchange", ShardedMsm(ctx, per, 0, 1, gather_device="cuda", force_exchange=False))
for gprio, pgprio in ((0, False), (1, False), (0, True), (1, True)):
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(so.getsockname()[1])
    sharding.PartialGatherer.STREAM_PRIORITY = -1 if gprio else 0
    opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=pgprio)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), pg_options=opts)
    run("one-rank RCCL: gatherer stream %s, RCCL stream %s" % ("high" if gprio else "normal", "high" if pgprio else "normal"),
        ShardedMsm(ctx, per, 0, 1, gather_device="cuda", force_exchange=True))
    dist.destroy_process_group()
