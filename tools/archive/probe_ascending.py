"""Per-call times of lone commitments with ascending sizes on one SRS (the 2^18 outlier of tools/time_commit_sizes.py)."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << 20, ctx=ctx)
out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
for log_n in range(15, 21):
    n = 1 << log_n
    d = torch.from_numpy(bench.blob_like_scalars(n, 123).view(np.int64)).cuda(); torch.cuda.synchronize()
    ts = []
    for _ in range(35):
        t0 = time.perf_counter()
        assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, 0) == 0
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, _lib.ptr(out), C.byref(inf), None) == 0
        ts.append((time.perf_counter() - t0) * 1e3)
    print("2^%d" % log_n, " ".join("%.2f" % t for t in ts), flush=True)
