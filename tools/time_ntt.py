"""ms per device-resident Fr NTT / INTT at a range of sizes (200 calls each, HIP-synchronised).  Usage (GPU box): [KZG_LIB_PATH=variant.so] python tools/time_ntt.py [log sizes]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
logs = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [12, 16, 18, 19, 20, 21, 22, 24]
row = []
for lg in logs:
    n = 1 << lg
    d = torch.zeros((n, 4), dtype=torch.int64, device="cuda"); d[:, 0] = torch.arange(n, device="cuda")
    out = []
    for inv in (0, 1):
        for _ in range(5):
            assert lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d.data_ptr()), n, inv) == 0
        torch.cuda.synchronize()
        reps = 200 if lg <= 22 else 40
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d.data_ptr()), n, inv)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / reps * 1e3)
        out.append(best)
    row.append("2^%d %.4f / %.4f" % (lg, out[0], out[1]))
print("NTT / INTT ms:", "  ".join(row), flush=True)
