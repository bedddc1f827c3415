"""ms per 2^LOG Fr NTT (device-resident), forward and inverse."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
for log_n in [int(a) for a in sys.argv[1:]] or [16, 18, 19, 20, 22]:
    n = 1 << log_n
    a = np.random.default_rng(1).integers(0, 1 << 60, size=(n, 4), dtype=np.uint64)
    d = torch.from_numpy(a.view(np.int64)).cuda(); torch.cuda.synchronize()
    for inv in (0, 1):
        for _ in range(5): lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d.data_ptr()), n, inv)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d.data_ptr()), n, inv)
        torch.cuda.synchronize()
        print("2^%d %s %.4f ms" % (log_n, "intt" if inv else "ntt ", (time.perf_counter() - t) / 50 * 1e3), flush=True)
