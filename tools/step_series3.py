"""Blocks of 20 pipelined steps, alternating the blob-like (A) and uniform (B) scalars of bench.py, after bench.py's own set-up."""
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
seed = 0x4B5A472D424E3235 & 0x7FFFFFFF
dA = torch.from_numpy(bench.ints_to_wire(bench.blob_like_canonical(n, seed)).view(np.int64)).cuda()
dB = torch.from_numpy(bench.uniform_scalars(n, seed + 1)[1].view(np.int64)).cuda()
torch.cuda.synchronize()
sh = ShardedMsm(ctx, n)
def block(d, cnt):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in sh.commit_stream(srs, [d.data_ptr()] * cnt, depth=2): pass
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / cnt * 1e3
block(dA, 2); block(dA, 48); block(dA, 5)
print(" ".join("%s %.3f" % (name, block(d, 20)) for name, d in [("A", dA), ("B", dB), ("A", dA), ("B", dB), ("A", dA), ("A", dA), ("B", dB), ("B", dB)]))
