"""Latency of one commitment of n coefficients (device-resident scalars) against ONE loaded SRS of 2^SRS_LOG points, n = 2^11 .. 2^SRS_LOG:
the table window bits c are fixed by the SRS size, so this shows what a given c costs the small blobs."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs_log = int(os.environ.get("SRS_LOG", "19"))
srs = k.SRS.generate(tau, 1 << srs_log, ctx=ctx)
out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
row = []
for log_n in range(11, srs_log + 1):
    n = 1 << log_n
    d = torch.from_numpy(bench.blob_like_scalars(n, 123).view(np.int64)).cuda(); torch.cuda.synchronize()
    def one():
        assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, 0) == 0
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, _lib.ptr(out), C.byref(inf), None) == 0
    for _ in range(5): one()
    t0 = time.perf_counter()
    for _ in range(30): one()
    row.append("2^%d %.3f" % (log_n, (time.perf_counter() - t0) / 30 * 1e3))
print("SRS 2^%d c=%s | ms per commitment: " % (srs_log, "auto") + "  ".join(row), flush=True)
