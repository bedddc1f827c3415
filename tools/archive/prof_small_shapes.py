"""The reference's small bench shapes for rocprofv3 --kernel-trace: 60 proofs at a domain point and 60 coefficient-form commitments of N (default 512)
elements from host buffers on a 2^15-point SRS."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
n = int(os.environ.get("N", "512"))
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << 15, ctx=ctx)
sc = bench.blob_like_scalars(n, 5)
kz = k.KZG.new(ctx); kz.calculate_and_store_roots_of_unity(32 * n)
roots = np.ascontiguousarray(kz.get_roots_of_unities()); z = np.ascontiguousarray(roots[n // 3])
o8 = np.zeros(8, np.uint64); o4 = np.zeros(4, np.uint64); oi = C.c_uint8(0)
def proof():
    assert lib.kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(sc), n, _lib.ptr(roots), n, _lib.ptr(z), _lib.ptr(o8), C.byref(oi), _lib.ptr(o4)) == 0
def cc():
    assert lib.kzg_commit_coeff_form(ctx.handle, srs.handle, _lib.ptr(sc), n, _lib.ptr(o8), C.byref(oi)) == 0
for f, name in ((proof, "compute_proof"), (cc, "commit_coeff_form")):
    for _ in range(5): f()
    t = time.perf_counter()
    for _ in range(60): f()
    print("%s n=%d %.4f ms" % (name, n, (time.perf_counter() - t) / 60 * 1e3), flush=True)
