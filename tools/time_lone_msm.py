"""Lone commitments at the reference's large-blob shapes (prover/benches/bench_kzg_commit_large_blobs.rs:17-37: 8 MiB = 2^18, 16 MiB = 2^19
coefficients): ONE commitment at a time over the loaded 2^20-point SRS --
  resident   kzg_msm_g1_srs_device_begin + _end, scalars already in HBM (median wall time of 40 calls + the library's per-phase HIP events)
  host       kzg_commit_coeff_form from a pageable host buffer (what the reference's bench hands over): + the H2D copy
Where the time of a lone MSM goes (device span from the events, the rest = host: enqueue before the first kernel, wake-up, epilogue).
Usage (GPU box): python tools/time_lone_msm.py [log sizes ...]     env KZG_* variants apply (tools/ab_env.sh)"""
import ctypes as C
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib

lib = _lib.load()
ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << 20, ctx=ctx)
out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
logs = [int(a) for a in sys.argv[1:]] or [17, 18, 19, 20]
names = ["digits", "sort1", "sort2", "-", "accumulate", "reduce1", "reduce2", "device span"]


def med(ts):
    ts = sorted(ts)
    return ts[len(ts) // 2]


for lg in logs:
    n = 1 << lg
    wire = bench.blob_like_scalars(n, 123)
    d = torch.from_numpy(wire.view(np.int64)).cuda()
    torch.cuda.synchronize()

    def resident():
        assert lib.kzg_msm_g1_srs_device_begin(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, 0) == 0
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, _lib.ptr(out), C.byref(inf), None) == 0

    def host():
        assert lib.kzg_commit_coeff_form(ctx.handle, srs.handle, _lib.ptr(wire), n, _lib.ptr(out), C.byref(inf)) == 0

    res = {}
    for name, fn in (("resident", resident), ("host", host)):
        for _ in range(6):
            fn()
        ts = []
        for _ in range(40):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        res[name] = (med(ts), min(ts))
    want = out.copy()
    lib.kzg_ctx_set_profiling(ctx.handle, 1)
    for _ in range(16):
        resident()
    phase = (C.c_double * 8)(); launches, pairs = C.c_uint64(0), C.c_uint64(0)
    lib.kzg_ctx_get_msm_profile(ctx.handle, phase, C.byref(launches), C.byref(pairs))
    lib.kzg_ctx_set_profiling(ctx.handle, 0)
    ph = [phase[i] / max(1, launches.value) for i in range(8)]
    assert np.array_equal(out, want)
    print("2^%d: resident median %.3f (min %.3f) ms, from a host buffer median %.3f (min %.3f) ms" % (lg, res["resident"][0], res["resident"][1], res["host"][0], res["host"][1]))
    print("      device: " + ", ".join("%s %.3f" % (nm, v) for nm, v in zip(names, ph) if nm != "-") + "; host side of a resident call: %.3f ms" % (res["resident"][0] - ph[7]), flush=True)
