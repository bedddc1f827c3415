"""bench_g1_ifft (prover/benches/bench_g1_ifft.rs:28-30: every power of two <= 2048) and larger sizes; cached-Lagrange commit."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << 16, ctx=ctx)
kz = k.KZG.new(ctx)
for log_n in [int(x) for x in os.environ.get("IFFT_LOGS", "0,1,2,3,4,5,6,7,8,9,10,11,12,14,16").split(",")]:
    n = 1 << log_n
    kz.g1_ifft(n, srs)
    reps = 5 if log_n <= 12 else 2
    t0 = time.perf_counter()
    for _ in range(reps):
        kz.g1_ifft(n, srs)
    print("g1_ifft(%6d): %8.3f ms" % (n, (time.perf_counter() - t0) / reps * 1e3), flush=True)
n = 1 << 16
ev = bench.ints_to_wire(bench.uniform_scalars(n, 5)[0])
poly = k.PolynomialEvalForm(ev)
def t(fn, reps=10):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
a = t(lambda: kz.commit_eval_form(poly, srs))
t0 = time.perf_counter(); srs.cache_lagrange(n); build = (time.perf_counter() - t0) * 1e3
b = t(lambda: kz.commit_eval_form(poly, srs))
print("commit_eval_form 2^16 from host buffers: IFFT + MSM %.3f ms; cached Lagrange basis %.3f ms (built once in %.1f ms)" % (a, b, build))
