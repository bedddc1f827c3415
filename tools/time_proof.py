"""compute_proof at 2^20 from host buffers: wall per call for several KZG_POLY_PER_LANE values (child processes)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes as C, hashlib, os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
n = 1 << 20
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") %% bench.FR
srs = k.SRS.generate(tau, n, ctx=ctx)
sc = bench.blob_like_scalars(n, 5)
o8 = np.zeros(8, np.uint64); o4 = np.zeros(4, np.uint64); oi = C.c_uint8(0)
zq = np.ascontiguousarray(sc[777])
def proof(): 
    assert lib.kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(sc), n, None, n, _lib.ptr(zq), _lib.ptr(o8), C.byref(oi), _lib.ptr(o4)) == 0
def ev():
    assert lib.kzg_evaluate_polynomial_in_evaluation_form(ctx.handle, _lib.ptr(sc), n, _lib.ptr(zq), _lib.ptr(o4)) == 0
for f, name in ((proof, "compute_proof"), (ev, "evaluate")):
    for _ in range(2): f()
    t = time.perf_counter()
    for _ in range(8): f()
    print("per_lane=%%s %%s %%.3f ms" %% (os.environ.get("KZG_POLY_PER_LANE", "32"), name, (time.perf_counter() - t) / 8 * 1e3), flush=True)
''' % ROOT
for v in os.environ.get("SWEEP", "32,16,8,4").split(","):
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, KZG_POLY_PER_LANE=v), check=False)
