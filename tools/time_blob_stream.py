"""ms per 32 MiB blob of the blob -> commitment + proof stream (kzg_commit_and_prove_blob_begin / _end) against the one-call entry.
Usage (GPU box): python tools/time_blob_stream.py [log_n=20] [blobs=32] [inflight list, default 1,2,4,6,8,12,16]"""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
count = int(sys.argv[2]) if len(sys.argv) > 2 else 32
depths = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 6, 8, 12, 16]
n = 1 << log_n
FR = 21888242871839275222246405745257275088548364400416034343698204186575808495617
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % FR
srs = k.SRS.generate(tau, n, ctx=ctx)
t0 = time.perf_counter(); srs.cache_lagrange(n); print("cache_lagrange %.2f s" % (time.perf_counter() - t0), flush=True)
rng = np.random.default_rng(7)
blobs = []
for _ in range(4):
    raw = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); raw[:, 0] &= 0x1F
    blobs.append(np.ascontiguousarray(raw.reshape(-1)))
u8p = C.POINTER(C.c_uint8)
def outs():
    return np.zeros(8, np.uint64), C.c_uint8(0), np.zeros(8, np.uint64), C.c_uint8(0), np.zeros(4, np.uint64), np.zeros(4, np.uint64)
def one_call(b):
    c, ci, p, pi, z, y = outs()
    assert lib.kzg_commit_and_prove_blob(ctx.handle, srs.handle, b.ctypes.data_as(u8p), b.size, n, _lib.ptr(c), C.byref(ci), _lib.ptr(p), C.byref(pi), _lib.ptr(z), _lib.ptr(y)) == 0
    return c, p, z, y
want = [one_call(b) for b in blobs]
t0 = time.perf_counter()
for i in range(8): one_call(blobs[i % 4])
print("one call at a time: %.2f ms per blob" % ((time.perf_counter() - t0) * 1e3 / 8), flush=True)
def begin(i, job):
    b = blobs[i % 4]
    rc = lib.kzg_commit_and_prove_blob_begin(ctx.handle, srs.handle, b.ctypes.data_as(u8p), b.size, n, None, job)
    assert rc == 0, (rc, ctx.last_error())
def end(i, job):
    c, ci, p, pi, z, y = outs()
    rc = lib.kzg_commit_and_prove_blob_end(ctx.handle, job, _lib.ptr(c), C.byref(ci), _lib.ptr(p), C.byref(pi), _lib.ptr(z), _lib.ptr(y))
    assert rc == 0, (rc, ctx.last_error())
    w = want[i % 4]
    assert np.array_equal(c, w[0]) and np.array_equal(p, w[1]) and np.array_equal(z, w[2]) and np.array_equal(y, w[3]), "stream result differs from the one-call entry"
def stream(depth, total):
    t0 = time.perf_counter()
    for i in range(total + depth):
        if i >= depth: end(i - depth, (i - depth) % depth)
        if i < total: begin(i, i % depth)
    return (time.perf_counter() - t0) * 1e3 / total
for d in depths:
    stream(d, max(d, 4))
    ts = [stream(d, count) for _ in range(3)]
    print("in flight %2d: %s ms per blob (min %.2f)" % (d, " ".join("%.2f" % t for t in ts), min(ts)), flush=True)
