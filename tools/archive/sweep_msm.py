"""Sweep MSM tunables on the GPU (segment length, table window bits): prints per-phase milliseconds.
Usage: python tools/sweep_msm.py [log_n]      env: SWEEP_C=16,15  SWEEP_L=16,32,64"""
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

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
lib = _lib.load()
ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
scalars = bench.blob_like_scalars(n, 12345)
rng = np.random.default_rng(5)
uni = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
uni[:, 3] &= np.uint64((1 << 60) - 1)
names = ["digits", "hist+scan", "scatter", "segments", "accumulate", "bucket_fin", "reduce", "device_total"]

for table_c in [int(x) for x in os.environ.get("SWEEP_C", "16").split(",")]:
    os.environ["KZG_TABLE_C"] = str(table_c)
    t0 = time.perf_counter()
    srs = k.SRS.generate(tau, n, ctx=ctx)
    t_srs = time.perf_counter() - t0
    for label, sc in (("blob", scalars), ("uniform", uni)):
        d = torch.from_numpy(sc.view(np.int64)).cuda()
        for L, mult in [(int(x), int(t)) for x in os.environ.get("SWEEP_L", "16,32,48,64,96,128").split(",")
                        for t in os.environ.get("SWEEP_TILE", "8").split(",")]:
            os.environ["KZG_SORT_TILE_MULT"] = str(mult)
            ctx.set_msm_window(0, L)
            out = np.zeros(8, np.uint64)
            inf = C.c_uint8(0)
            for it in range(2):
                lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(out), C.byref(inf))
            lib.kzg_ctx_set_profiling(ctx.handle, 1)
            t0 = time.perf_counter()
            reps = 5
            for it in range(reps):
                lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d.data_ptr()), n, _lib.ptr(out), C.byref(inf))
            wall = (time.perf_counter() - t0) / reps * 1e3
            ph = (C.c_double * 8)()
            la = C.c_uint64(0)
            pa = C.c_uint64(0)
            lib.kzg_ctx_get_msm_profile(ctx.handle, ph, C.byref(la), C.byref(pa))
            lib.kzg_ctx_set_profiling(ctx.handle, 0)
            print(f"c={table_c} (srs {t_srs*1e3:.0f} ms) {label:8s} L={L:4d} tile={mult:2d} wall={wall:6.3f} ms | " +
                  " ".join(f"{nm}={ph[i]/reps:.3f}" for i, nm in enumerate(names)), flush=True)
    srs.close()
