"""Where does a pass of the Fr NTT spend its time: per-workgroup phase sums (s_memrealtime, 100 MHz) from a diagnostic build
(tools/build_variant.sh nttstamps -DKZG_NTT_STAMPS; KZG_LIB_PATH=gpurun_variants/libkzg_nttstamps.so python tools/ntt_stamps.py)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
raw = C.CDLL(_lib.LIB_PATH)
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
a = np.random.default_rng(1).integers(0, 1 << 60, size=(n, 4), dtype=np.uint64)
d = torch.from_numpy(a.view(np.int64)).cuda(); torch.cuda.synchronize()
for _ in range(20): lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d.data_ptr()), n, 0)
torch.cuda.synchronize()
out = np.zeros((4, 1024, 8), np.uint64)
assert raw.kzg_debug_ntt_stamps(out.ctypes.data_as(C.c_void_p)) == 0
names = ["setup / trailing barrier", "wait words + unpack + fill", "prefetch issue", "radix-4 steps", "odd radix-2 stage", "drain: LDS read + multiply + pack + stores", "exit"]
for p in range(2):
    wg = out[p][:256].astype(np.float64) * 10.0 / 1000.0          # ticks of 10 ns -> us
    tot = wg.sum(axis=1)
    print("pass %d: per workgroup total %.1f us (min %.1f max %.1f)" % (p, tot.mean(), tot.min(), tot.max()))
    for i, nm in enumerate(names):
        print("   %-44s %7.2f us  %5.1f %%" % (nm, wg[:, i].mean(), 100 * wg[:, i].mean() / tot.mean()))
