"""Latency of commit_eval_form / compute_proof (z off the domain, z on the domain) / NTT over sizes 2^1 .. 2^17: look for outliers."""
import ctypes as C, hashlib, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import pyref
from pyref import R_
import rust_kzg_bn254_amd as k
lib = k._lib.load(); ctx = k.Context(0)
TAU = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % R_
MONT = (1 << 256) % R_
rnd = random.Random(1)
def wire(vals): return np.frombuffer(b"".join((v * MONT % R_).to_bytes(32, "little") for v in vals), dtype=np.uint64).reshape(-1, 4).copy()
srs = k.SRS.generate(TAU, 1 << 17, ctx=ctx)
def t3(fn):
    fn(); fn()
    t0 = time.perf_counter(); fn(); fn(); fn()
    return (time.perf_counter() - t0) / 3 * 1e3
for log_n in range(1, 18):
    n = 1 << log_n
    ev = wire([rnd.randrange(R_) for _ in range(n)])
    out = np.zeros(8, np.uint64); inf = C.c_uint8(0); y = np.zeros(4, np.uint64)
    z_off = wire([rnd.randrange(R_)])[0].copy()
    w = pyref.root_of_unity(log_n)
    z_on = wire([pow(w, 3 % n, R_)])[0].copy()
    a = t3(lambda: lib.kzg_commit_eval_form(ctx.handle, srs.handle, k._lib.ptr(ev), n, k._lib.ptr(out), C.byref(inf)))
    b = t3(lambda: lib.kzg_compute_proof(ctx.handle, srs.handle, k._lib.ptr(ev), n, None, n, k._lib.ptr(z_off), k._lib.ptr(out), C.byref(inf), k._lib.ptr(y)))
    c = t3(lambda: lib.kzg_compute_proof(ctx.handle, srs.handle, k._lib.ptr(ev), n, None, n, k._lib.ptr(z_on), k._lib.ptr(out), C.byref(inf), k._lib.ptr(y)))
    d = t3(lambda: lib.kzg_fr_ntt(ctx.handle, k._lib.ptr(ev), n, 0))
    print("2^%-2d commit_eval %.3f  proof(off) %.3f  proof(on) %.3f  ntt(host buffers) %.3f" % (log_n, a, b, c, d), flush=True)
