"""Host cost of one MSM outside the device work: time of kzg_msm_g1_srs_device_begin (launches) and of kzg_msm_g1_srs_end called
after the device has long finished (event wait that returns at once + host epilogue)."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
for log_n in [int(x) for x in os.environ.get("LOGS", "11,17,20").split(",")]:
    n = 1 << log_n
    srs = k.SRS.generate(tau, max(n, 1 << int(os.environ.get("SRS_LOG", "0"))), ctx=ctx)
    d = torch.from_numpy(bench.blob_like_scalars(n, 123).view(np.int64)).cuda(); torch.cuda.synchronize()
    out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
    tb = te = tw = 0.0
    reps = 20
    for r in range(reps + 3):
        t0 = time.perf_counter()
        assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, 0) == 0
        t1 = time.perf_counter()
        time.sleep(0.01)
        t2 = time.perf_counter()
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, _lib.ptr(out), C.byref(inf), None) == 0
        t3 = time.perf_counter()
        if r >= 3: tb += t1 - t0; te += t3 - t2
    # whole synchronous call
    for r in range(reps + 3):
        t0 = time.perf_counter()
        assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, 0) == 0
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, _lib.ptr(out), C.byref(inf), None) == 0
        if r >= 3: tw += time.perf_counter() - t0
    print("n=2^%d (SRS %d): begin %.1f us, end after idle %.1f us, begin+end back to back %.1f us" % (log_n, srs.len if hasattr(srs, "len") else n, tb / reps * 1e6, te / reps * 1e6, tw / reps * 1e6), flush=True)
    srs.close()
