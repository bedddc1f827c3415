"""Correctness + time of one commitment at 2^LOG (default 21: beyond the 24-bit index of the two-level sort) on the known-tau SRS."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bench, pyref
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
log_n = int(os.environ.get("LOG", "21")); n = 1 << log_n
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
t0 = time.perf_counter(); srs = k.SRS.generate(tau, n, ctx=ctx); print("SRS 2^%d + tables: %.1f ms" % (log_n, (time.perf_counter() - t0) * 1e3), flush=True)
sc = bench.blob_like_scalars(n, 11)
d = torch.from_numpy(sc.view(np.int64)).cuda(); torch.cuda.synchronize()
out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
def one():
    assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, 0) == 0
    assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, _lib.ptr(out), C.byref(inf), None) == 0
for _ in range(3): one()
t0 = time.perf_counter()
for _ in range(10): one()
print("commit 2^%d: %.3f ms" % (log_n, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
want = bench.expected_commitment(bench.blob_like_canonical(n, 11), tau)
print("matches p(tau)*G1:", np.array_equal(np.asarray(want, dtype=np.uint64).reshape(-1), out), flush=True)
