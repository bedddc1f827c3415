"""Times kzg_fr_ntt_device (forward / inverse) on device-resident data for several sizes; prints per-pass rocprof-free wall averages."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
for log_n in (16, 18, 20, 22, 24):
    n = 1 << log_n
    rng = np.random.default_rng(1)
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
    d = torch.from_numpy(a.view(np.int64)).cuda(); torch.cuda.synchronize()
    for inv in (0, 1):
        for _ in range(3):
            assert lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d.data_ptr()), n, inv) == 0
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d.data_ptr()), n, inv)
        dt = (time.perf_counter() - t0) / reps
        print(f"n=2^{log_n} inverse={inv}: {dt*1e3:.4f} ms  ({64*n/dt/1e9:.0f} GB/s algorithmic)", flush=True)
