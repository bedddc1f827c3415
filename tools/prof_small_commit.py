"""20 synchronous coefficient-form commitments of n points (argv[1], default 2048) for rocprofv3 --kernel-trace."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rust_kzg_bn254_amd as k
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
TAU = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % k.consts.FR_MODULUS
srs = k.SRS.generate(TAU, 1 << 19)
kz = k.KZG.new()
rng = np.random.default_rng(1)
poly = k.Blob.from_raw_data(rng.integers(32, 127, size=n * 31 - 5, dtype=np.uint8).tobytes()).to_polynomial_coeff_form()
for _ in range(5): kz.commit_coeff_form(poly, srs)
t = time.perf_counter()
for _ in range(20): kz.commit_coeff_form(poly, srs)
print("n =", len(poly), "ms per commit", (time.perf_counter() - t) / 20 * 1e3)
