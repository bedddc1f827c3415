"""Phase times of kzg_verify_blob_kzg_proof_batch (BASELINE config 5, 4 096 rows) with KZG_VB_TRACE=1: the bench's workload (256 distinct blobs of
35 .. 50 000 raw bytes, each used 16 times), 12 calls.  Usage (GPU box): KZG_VB_TRACE=1 python tools/trace_batch_verify.py"""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
from rust_kzg_bn254_amd.helpers import pad_payload
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << 12, ctx=ctx)
u8p = C.POINTER(C.c_uint8)
rng5 = np.random.default_rng(5)
rows5 = []
for n_raw in rng5.integers(35, 50000, size=256):
    data = pad_payload(rng5.integers(32, 127, size=int(n_raw), dtype=np.uint8).tobytes())
    npad = 1
    while npad < len(data) // 32:
        npad <<= 1
    buf = np.frombuffer(data, dtype=np.uint8)
    c5 = np.zeros(8, np.uint64); p5 = np.zeros(8, np.uint64); ci5 = C.c_uint8(0); pi5 = C.c_uint8(0)
    assert lib.kzg_commit_and_prove_blob(ctx.handle, srs.handle, buf.ctypes.data_as(u8p), len(data), npad, _lib.ptr(c5), C.byref(ci5), _lib.ptr(p5), C.byref(pi5), None, None) == 0
    rows5.append((data, c5, p5))
nb = 4096
sel5 = [rows5[i % 256] for i in range(nb)]
ptrs5, lens5, _keep5 = _lib.blob_args([r[0] for r in sel5])
cm5 = np.ascontiguousarray(np.stack([r[1] for r in sel5])); pf5 = np.ascontiguousarray(np.stack([r[2] for r in sel5]))
tau_g2 = np.zeros(16, np.uint64)
lib.kzg_g2_mul_generator(_lib.ptr(k.fr.fr_from_int(tau)), _lib.ptr(tau_g2))
ok5 = C.c_int32(0)
ts = []
for i in range(12):
    t0 = time.perf_counter()
    assert lib.kzg_verify_blob_kzg_proof_batch(ctx.handle, ptrs5, lens5, _lib.ptr(cm5), _lib.ptr(pf5), nb, _lib.ptr(tau_g2), C.byref(ok5)) == 0 and ok5.value == 1
    ts.append((time.perf_counter() - t0) * 1e3)
print("per call ms:", " ".join("%.2f" % t for t in ts), "| blob MiB", sum(len(r[0]) for r in sel5) / 2 ** 20, flush=True)
