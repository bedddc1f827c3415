"""One-call entry points at 2^LOG_N from a PAGEABLE host buffer (what a Rust Vec<Fr> is) against the same buffer PINNED (hipHostRegister through
torch's pin_memory): what a host that registers its buffers once would gain.  Usage (GPU box): python tools/probe_pinned_host.py"""
import ctypes as C
import hashlib
import os
import statistics
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
P = _lib.ptr
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
for log_n in [int(x) for x in os.environ.get("LOG_NS", "18,19,20").split(",")]:
    n = 1 << log_n
    srs = k.SRS.generate(tau, n, ctx=ctx)
    assert lib.kzg_srs_cache_lagrange(ctx.handle, srs.handle, n) == 0
    sc = bench.blob_like_scalars(n, 5)
    pinned = torch.from_numpy(sc.view(np.int64)).pin_memory()
    o8 = np.zeros(8, np.uint64); o4 = np.zeros(4, np.uint64); oi = C.c_uint8(0)
    z = np.ascontiguousarray(bench.uniform_scalars(4, 99)[1][1])
    u64p = C.POINTER(C.c_uint64)
    bufs = {"pageable": P(sc), "pinned": C.cast(C.c_void_p(pinned.data_ptr()), u64p)}
    calls = {
        "commit_coeff_form": lambda p: lib.kzg_commit_coeff_form(ctx.handle, srs.handle, p, n, P(o8), C.byref(oi)),
        "commit_eval_form (cached basis)": lambda p: lib.kzg_commit_eval_form(ctx.handle, srs.handle, p, n, P(o8), C.byref(oi)),
        "compute_proof (cached basis)": lambda p: lib.kzg_compute_proof(ctx.handle, srs.handle, p, n, None, n, P(z), P(o8), C.byref(oi), P(o4)),
    }
    for name, f in calls.items():
        row = []
        want = None
        for kind, p in bufs.items():
            for _ in range(3):
                assert f(p) == 0
            ts = []
            for _ in range(30):
                t0 = time.perf_counter(); rc = f(p); ts.append(time.perf_counter() - t0)
                assert rc == 0
            want = o8.copy() if want is None else want
            assert np.array_equal(want, o8)
            row.append("%s %.3f ms" % (kind, statistics.median(ts) * 1e3))
        print("2^%d %-34s %s" % (log_n, name, " | ".join(row)), flush=True)
    srs.close()
