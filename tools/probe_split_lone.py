"""Would a LONE large MSM finish sooner as two halves in flight on two slots (the second half's sort beside the first half's accumulate) + a host
fold of the two partial sums?  Resident scalars, median of 30.  Usage (GPU box): python tools/probe_split_lone.py"""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0); P = _lib.ptr
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << 20, ctx=ctx)
out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
def med(fn, reps=30):
    for _ in range(5): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]
for lg in (18, 19, 20):
    n = 1 << lg
    d = torch.from_numpy(bench.blob_like_scalars(n, 123).view(np.int64)).cuda(); torch.cuda.synchronize()
    base = d.data_ptr()
    def whole():
        assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, 0, C.c_void_p(base), n, 0) == 0
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, P(out), C.byref(inf), None) == 0
    want = None
    t_whole = med(whole); want = out.copy()
    res = {}
    for parts in (2, 3, 4):
        ln = n // parts
        bounds = [(i * ln, (i + 1) * ln if i < parts - 1 else n) for i in range(parts)]
        pp = np.zeros((parts, 16), np.uint64)
        def split():
            for i, (lo, hi) in enumerate(bounds):
                assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, lo, C.c_void_p(base + lo * 32), hi - lo, i) == 0
            for i in range(parts):
                assert lib.kzg_msm_g1_srs_end(ctx.handle, i, None, None, P(pp[i])) == 0
            assert lib.kzg_g1_fold_partials(P(pp), parts, P(out), C.byref(inf)) == 0
        res[parts] = med(split)
        assert np.array_equal(out, want), (lg, parts)
    print("2^%d: whole %.3f ms; as %s" % (lg, t_whole, ", ".join("%d parts %.3f" % (p_, v) for p_, v in res.items())), flush=True)
