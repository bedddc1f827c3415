"""Times the secondary paths on the GPU: Fr NTT / INTT (device-resident), commit_eval_form, compute_proof."""
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
rng = np.random.default_rng(5)
uni = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
uni[:, 3] &= np.uint64((1 << 60) - 1)
d = torch.from_numpy(uni.view(np.int64)).cuda()
torch.cuda.synchronize()


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


t_fwd = timeit(lambda: lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d.data_ptr()), n, 0))
t_inv = timeit(lambda: lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d.data_ptr()), n, 1))
print(f"NTT 2^{log_n}: fwd {t_fwd:.3f} ms  inv {t_inv:.3f} ms  -> {64*n/t_fwd/1e6:.1f} GB/s algorithmic (64 B/elem)", flush=True)

srs = k.SRS.generate(tau, n, ctx=ctx)
out = np.zeros(8, np.uint64); inf = C.c_uint8(0); y = np.zeros(4, np.uint64)
z = np.ascontiguousarray(uni[7])
t_ce = timeit(lambda: lib.kzg_commit_eval_form(ctx.handle, srs.handle, _lib.ptr(uni), n, _lib.ptr(out), C.byref(inf)), reps=5)
t_cc = timeit(lambda: lib.kzg_commit_coeff_form(ctx.handle, srs.handle, _lib.ptr(uni), n, _lib.ptr(out), C.byref(inf)), reps=5)
t_pr = timeit(lambda: lib.kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(uni), n, None, n, _lib.ptr(z), _lib.ptr(out), C.byref(inf), _lib.ptr(y)), reps=5)
t_ev = timeit(lambda: lib.kzg_evaluate_polynomial_in_evaluation_form(ctx.handle, _lib.ptr(uni), n, _lib.ptr(z), _lib.ptr(y)), reps=5)
print(f"host-buffer paths 2^{log_n}: commit_coeff {t_cc:.3f} ms  commit_eval {t_ce:.3f} ms  compute_proof {t_pr:.3f} ms  evaluate {t_ev:.3f} ms", flush=True)

kz = k.KZG.new(ctx)
for m in (64, 2048):
    t0 = time.perf_counter(); kz.g1_ifft(m, srs); t = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); kz.g1_ifft(m, srs); t2 = (time.perf_counter() - t0) * 1e3
    print(f"g1_ifft({m}): {t2:.2f} ms (first call {t:.2f} ms)", flush=True)
