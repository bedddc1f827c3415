"""A few forward NTTs at one size (argv[1] = log2 n) for rocprofv3 --kernel-trace."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
lib = _lib.load(); ctx = k.Context(0)
a = np.random.default_rng(1).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
d = torch.from_numpy(a.view(np.int64)).cuda(); torch.cuda.synchronize()
for _ in range(10):
    assert lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d.data_ptr()), n, 0) == 0
