"""Streamed host-buffer commitments (kzg_msm_g1_srs_begin / _end, two in flight): pageable against pinned caller buffers."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
n = 1 << 20
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, n, ctx=ctx)
sets = [bench.blob_like_scalars(n, 100 + j) for j in range(4)]
pinned_t = [torch.from_numpy(s.view(np.int64)).pin_memory() for s in sets]
pinned = [t.numpy().view(np.uint64) for t in pinned_t]
o8 = np.zeros(8, np.uint64); oi = C.c_uint8(0)
def stream(bufs, reps, depth=2):
    inflight = []
    for i in range(reps):
        if len(inflight) == depth:
            assert lib.kzg_msm_g1_srs_end(ctx.handle, inflight.pop(0), _lib.ptr(o8), C.byref(oi), None) == 0
        slot = i % depth
        assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, 0, _lib.ptr(bufs[i % len(bufs)]), n, slot) == 0
        inflight.append(slot)
    while inflight:
        assert lib.kzg_msm_g1_srs_end(ctx.handle, inflight.pop(0), _lib.ptr(o8), C.byref(oi), None) == 0
import gc; gc.disable()
for name, bufs in (("pageable", sets), ("pinned", pinned)):
    for depth in (2, 3):
        stream(bufs, 8, depth)
        t = time.perf_counter(); stream(bufs, 48, depth); dt = (time.perf_counter() - t) / 48 * 1e3
        print("%s buffers, depth %d: %.3f ms per 2^20 commitment (%.1f GB/s of scalars)" % (name, depth, dt, n * 32 / dt / 1e6), flush=True)
