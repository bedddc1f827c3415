"""SRS.new (read gnark-format compressed G1 file, decompress on the GPU, build window tables) at 2^20 points.
The file is the reference's 3000-point fixture repeated (decompression cost does not depend on the values)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rust_kzg_bn254_amd as k
raw = open(os.path.join(ROOT, "tests", "golden", "g1.point"), "rb").read()
n = 1 << 20
data = (raw * (n * 32 // len(raw) + 1))[: n * 32]
with tempfile.NamedTemporaryFile(suffix=".point", delete=False) as f:
    f.write(data); path = f.name
k.default_context()
for it in range(3):
    t0 = time.perf_counter()
    srs = k.SRS.new(path, n, n)
    dt = time.perf_counter() - t0
    print(f"SRS.new 2^20 points (32 MiB file): {dt*1e3:.1f} ms", flush=True)
    srs.close()
os.unlink(path)
