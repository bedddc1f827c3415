"""Diagnostic: per-wave start / end time and placement of k_msm_accumulate (library variant built with -DKZG_ACC_STAMPS)."""
import ctypes as C, hashlib, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("KZG_LIB_PATH", os.path.join(ROOT, "gpurun_variants", "libkzg_stamps.so"))
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
n = 1 << int(os.environ.get("LOG_N", "20"))
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << 20, ctx=ctx)
d = torch.from_numpy(bench.blob_like_scalars(n, 12345).view(np.int64)).cuda(); torch.cuda.synchronize()
out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
for it in range(6):
    lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(out), C.byref(inf))
nw = int(os.environ.get("NW", "3072"))
st = np.zeros((nw, 8), np.uint64)
dbg = C.CDLL(os.environ["KZG_LIB_PATH"]).kzg_debug_acc_stamps
dbg.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
assert dbg(ctx.handle, st.ctypes.data, nw, nw * 64) == 0
t0 = st[:, 0].min()
start = (st[:, 0] - t0).astype(np.float64) / 100.0      # us
end = (st[:, 1] - t0).astype(np.float64) / 100.0
hw = st[:, 2].astype(np.uint32); xcc = (st[:, 3] & 0xF).astype(np.int64)
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 3
print("waves", nw, "start us: min %.1f max %.1f   end us: min %.1f mean %.1f max %.1f" % (start.min(), start.max(), end.min(), end.mean(), end.max()))
print("end-time percentiles (us):", np.percentile(end, [1, 10, 25, 50, 75, 90, 99]).round(1))
for x in range(8):
    m = xcc == x
    print("XCC %d: waves %4d  end mean %.1f  max %.1f   start max %.1f" % (x, m.sum(), end[m].mean(), end[m].max(), start[m].max()))
key = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
per = collections.Counter(key.tolist())
print("waves per (xcc,se,sh,cu,simd): distinct SIMDs %d, histogram of waves/SIMD %s" % (len(per), collections.Counter(per.values())))
last = collections.defaultdict(float)
for kk, e in zip(key.tolist(), end.tolist()):
    last[kk] = max(last[kk], e)
l = np.array(list(last.values()))
print("per-SIMD finish time (us): min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % (l.min(), np.percentile(l, 10), np.median(l), np.percentile(l, 90), l.max()))
# waves sorted by id: first and last few
order = np.argsort(end)
print("earliest finishing waves:", [(int(i), round(float(end[i]), 1)) for i in order[:6]])
print("latest finishing waves:", [(int(i), round(float(end[i]), 1), int(xcc[i])) for i in order[-10:]])
dur = end - start
print("wave duration us: min %.1f median %.1f max %.1f" % (dur.min(), np.median(dur), dur.max()))

ft = st[:, 4].astype(np.float64) / 100.0; fe = st[:, 5].astype(np.float64); pro = (st[:, 6] - st[:, 0]).astype(np.float64) / 100.0
print("prologue (search) us: median %.1f max %.1f" % (np.median(pro), pro.max()))
print("flush events per wave: mean %.1f max %d; flush time per wave us: mean %.1f max %.1f; per event us: %.2f" % (fe.mean(), fe.max(), ft.mean(), ft.max(), ft.sum() / max(fe.sum(), 1)))
print("corr(duration, flush time) = %.2f" % np.corrcoef(dur, ft)[0, 1])
