"""Workload for `rocprofv3 --kernel-trace --stats`: g1_ifft at the sizes in IFFT_LOGS (default 9,10,11), 10 calls each, on a 2^16-point SRS."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import bench
import rust_kzg_bn254_amd as k
ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << 16, ctx=ctx)
kz = k.KZG.new(ctx)
for log_n in [int(x) for x in os.environ.get("IFFT_LOGS", "9,10,11").split(",")]:
    for _ in range(10):
        kz.g1_ifft(1 << log_n, srs)
