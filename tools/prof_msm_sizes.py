"""Runs a few unpipelined MSMs at one size (argv[1] = log2 n, default 20) for rocprofv3 --kernel-trace --stats."""
import ctypes as C, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, n, ctx=ctx)
d = torch.from_numpy(bench.blob_like_scalars(n, 12345).view(np.int64)).cuda(); torch.cuda.synchronize()
out = np.zeros(16, np.uint64)
for it in range(20):
    assert lib.kzg_msm_g1_srs_partial_device(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(out)) == 0
