"""Device-resident Fr NTT timings (median of 200 forward + inverse transforms each) for the library selected by KZG_LIB_PATH."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
row = []
for log_n in [int(v) for v in os.environ.get("NTT_LOGS", "12,16,18,20,22,24").split(",")]:
    n = 1 << log_n
    d = torch.randint(0, 1 << 60, (n, 4), dtype=torch.int64, device="cuda")
    for inv in (0, 1):
        for _ in range(10): lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d.data_ptr()), n, inv)
    torch.cuda.synchronize()
    ts = []
    reps = 200 if log_n <= 20 else 40
    for i in range(reps):
        t0 = time.perf_counter(); assert lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d.data_ptr()), n, i & 1) == 0; torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    row.append("2^%d %.4f" % (log_n, ts[len(ts) // 2] * 1e3))
print("%s | median ms: " % os.path.basename(os.environ.get("KZG_LIB_PATH", "default")) + "  ".join(row), flush=True)
