"""Randomised soak of the batched commitments (kzg_commit_coeff_form_batch) against single calls: random lengths (every bucket width),
counts (one launch, several launches, ragged tails) and scalar shapes.  SOAK_SECONDS (default 60)."""
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
N = 1 << 16
srs = k.SRS.generate(TAU, N, ctx=ctx)
seed = int(os.environ.get("SOAK_SEED", str(int(time.time()))))
rnd = random.Random(seed)
print("seed", seed, flush=True)
def wire(vals): return np.frombuffer(b"".join((v * MONT % R_).to_bytes(32, "little") for v in vals), dtype=np.uint64).reshape(-1, 4).copy()
def poly(n):
    kind = rnd.randrange(5)
    if kind == 0: return [rnd.randrange(R_) for _ in range(n)]
    if kind == 1: return [rnd.randrange(1 << rnd.choice((8, 40, 130, 248))) for _ in range(n)]
    if kind == 2:
        few = [rnd.randrange(R_) for _ in range(3)] + [0, 1, R_ - 1]
        return [rnd.choice(few) for _ in range(n)]
    if kind == 3: return [0] * n
    v = rnd.randrange(R_); return [v] * n
t_end = time.time() + float(os.environ.get("SOAK_SECONDS", "60"))
cases = 0
while time.time() < t_end:
    n = rnd.choice((1, 2, 63, 64, 65, 500, 512, 1000, 2048, 4096, 8191, 8192, 8193, 16384, 20000, 32768, 40000, 65536))
    cap = int(lib.kzg_msm_batch_capacity(n))
    count = rnd.choice((1, 2, 3, cap, cap + 1, 2 * cap + 1)) if n >= 8192 else rnd.choice((1, 2, 7, 64, 200))
    count = max(1, min(count, (1 << 21) // n + 1))
    base = [poly(n) for _ in range(min(count, 6))]
    polys = [base[j % len(base)] for j in range(count)]
    data = np.concatenate([wire(p) for p in polys])
    out = np.zeros((count, 8), np.uint64); infs = np.zeros(count, np.uint8)
    rc = lib.kzg_commit_coeff_form_batch(ctx.handle, srs.handle, k._lib.ptr(data), n, count, k._lib.ptr(out), infs.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert rc == 0, (rc, n, count)
    for j in range(min(count, 6)):
        one = np.zeros(8, np.uint64); inf = C.c_uint8(0)
        assert lib.kzg_commit_coeff_form(ctx.handle, srs.handle, k._lib.ptr(wire(base[j])), n, k._lib.ptr(one), C.byref(inf)) == 0
        for q in range(j, count, len(base)):
            if not np.array_equal(out[q], one) or infs[q] != inf.value:
                print("MISMATCH seed", seed, "case", cases, "n", n, "count", count, "poly", q, flush=True); sys.exit(1)
    cases += 1
print("batch soak ok:", cases, "cases", flush=True)
