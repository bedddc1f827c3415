"""Randomised soak of the blob -> commitment + proof stream (kzg_commit_and_prove_blob_begin / _end) against the one-call entries: random blob
lengths (1 byte .. 2^SOAK_MAX_LOG elements, ragged tails, chunks >= r), random numbers of jobs in flight, random end order, some jobs with a given
commitment, a cached Lagrange basis for some sizes, now and then one of the caller's own asynchronous MSMs holding a slot.  SOAK_SECONDS (default
60), SOAK_SEED.  Usage (GPU box): python tools/soak_blob_stream.py"""
import ctypes as C, hashlib, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib

seconds = float(os.environ.get("SOAK_SECONDS", "60")); seed = int(os.environ.get("SOAK_SEED", str(int(time.time()))))
max_log = int(os.environ.get("SOAK_MAX_LOG", "16"))
rnd = random.Random(seed)
FR = 21888242871839275222246405745257275088548364400416034343698204186575808495617
lib = _lib.load(); ctx = k.Context(0)
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % FR
srs = k.SRS.generate(tau, 1 << max_log, ctx=ctx)
for lg in (10, 13, max_log):
    srs.cache_lagrange(1 << lg)
u8p = C.POINTER(C.c_uint8)


def padded(n_bytes):
    e, p = (n_bytes + 31) // 32, 1
    while p < e:
        p <<= 1
    return p


def one_call(buf, given=None):
    c = np.zeros(8, np.uint64); p = np.zeros(8, np.uint64); z = np.zeros(4, np.uint64); y = np.zeros(4, np.uint64); ci = C.c_uint8(0); pi = C.c_uint8(0)
    if given is None:
        rc = lib.kzg_commit_and_prove_blob(ctx.handle, srs.handle, buf.ctypes.data_as(u8p), buf.size, padded(buf.size), _lib.ptr(c), C.byref(ci), _lib.ptr(p), C.byref(pi), _lib.ptr(z), _lib.ptr(y))
    else:
        c[:] = given
        rc = lib.kzg_compute_blob_proof(ctx.handle, srs.handle, buf.ctypes.data_as(u8p), buf.size, padded(buf.size), _lib.ptr(c), _lib.ptr(p), C.byref(pi), _lib.ptr(z), _lib.ptr(y))
    assert rc == 0, rc
    return c, p, z, y


t_end = time.time() + seconds
rounds = jobs_done = 0
scal = np.frombuffer(np.random.default_rng(seed).bytes(32 * 4096), dtype=np.uint64).copy().reshape(4096, 4); scal[:, 3] &= (1 << 60) - 1
while time.time() < t_end:
    depth = rnd.choice([1, 2, 3, 4, 6, 8, 12, 16])
    blobs = []
    for _ in range(depth):
        lg = rnd.choice([0, 3, 7, 10, 12, 13, max_log - 1, max_log])
        n_bytes = max(1, 32 * (1 << lg) - rnd.choice([0, 0, 1, 17, 31, 32 * rnd.randrange(1 << lg) if lg else 0]))
        raw = np.frombuffer(bytes(rnd.getrandbits(8) for _ in range(min(n_bytes, 4096))) * (n_bytes // 4096 + 1), dtype=np.uint8)[:n_bytes].copy()
        if rnd.random() < 0.7:
            raw[::32] &= 0x1F                                  # mostly canonical chunks; the rest exercises the mod-r reduction of the transcript
        blobs.append(raw)
    wants = [one_call(b) for b in blobs]
    given = [rnd.random() < 0.3 for _ in blobs]
    hold = rnd.random() < 0.25
    if hold:                                                   # one of the caller's own asynchronous MSMs holds slot 2 meanwhile
        assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, 0, _lib.ptr(scal), 4096, 2) == 0
    order = list(range(depth))
    for j in order:
        cptr = _lib.ptr(wants[j][0]) if given[j] else None
        rc = lib.kzg_commit_and_prove_blob_begin(ctx.handle, srs.handle, blobs[j].ctypes.data_as(u8p), blobs[j].size, padded(blobs[j].size), cptr, j)
        assert rc == 0, (rc, ctx.last_error(), seed, rounds)
    rnd.shuffle(order)
    for j in order:
        c = np.zeros(8, np.uint64); p = np.zeros(8, np.uint64); z = np.zeros(4, np.uint64); y = np.zeros(4, np.uint64); ci = C.c_uint8(0); pi = C.c_uint8(0)
        rc = lib.kzg_commit_and_prove_blob_end(ctx.handle, j, _lib.ptr(c), C.byref(ci), _lib.ptr(p), C.byref(pi), _lib.ptr(z), _lib.ptr(y))
        assert rc == 0, (rc, ctx.last_error(), seed, rounds)
        w = wants[j]
        assert np.array_equal(c, w[0]) and np.array_equal(p, w[1]) and np.array_equal(z, w[2]) and np.array_equal(y, w[3]), ("MISMATCH", seed, rounds, j, blobs[j].size)
        jobs_done += 1
    if hold:
        o = np.zeros(8, np.uint64); inf = C.c_uint8(0)
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 2, _lib.ptr(o), C.byref(inf), None) == 0
    rounds += 1
print("blob stream soak ok: seed %d, %d rounds, %d jobs in %.0f s" % (seed, rounds, jobs_done, seconds), flush=True)
