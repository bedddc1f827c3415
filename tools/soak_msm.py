"""Randomised soak of the SRS MSM entry points against big-integer arithmetic on the known-tau SRS: random lengths, offsets, scalar
shapes (uniform, short, few distinct values, zeros, r - 1), synchronous and begin/end calls.  SOAK_SECONDS (default 60), SOAK_SRS_LOG (18)."""
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
log_srs = int(os.environ.get("SOAK_SRS_LOG", "18")); N = 1 << log_srs
srs = k.SRS.generate(TAU, N, ctx=ctx)
seed = int(os.environ.get("SOAK_SEED", str(int(time.time()))))
rnd = random.Random(seed)
print("seed", seed, "SRS 2^%d" % log_srs, flush=True)
def wire(vals): return np.frombuffer(b"".join((v * MONT % R_).to_bytes(32, "little") for v in vals), dtype=np.uint64).reshape(-1, 4).copy()
tp_cache = {}
def expect(sc, off):
    acc, tp = 0, pow(TAU, off, R_)
    for s in sc:
        acc = (acc + s * tp) % R_; tp = tp * TAU % R_
    return pyref.ec_mul(acc, (1, 2)) if acc else None
def scalars(n):
    kind = rnd.randrange(6)
    if kind == 0: return [rnd.randrange(R_) for _ in range(n)]
    if kind == 1: return [rnd.randrange(1 << rnd.choice((8, 40, 130, 248))) for _ in range(n)]
    if kind == 2:
        few = [rnd.randrange(R_) for _ in range(rnd.randrange(1, 6))] + [0, 1, R_ - 1]
        return [rnd.choice(few) for _ in range(n)]
    if kind == 3: return [rnd.randrange(R_) if rnd.random() < 0.1 else 0 for _ in range(n)]
    if kind == 4: return [R_ - 1 - rnd.randrange(1 << 20) for _ in range(n)]
    v = rnd.randrange(R_); return [v] * n
t_end = time.time() + float(os.environ.get("SOAK_SECONDS", "60"))
cases = 0
edges = [1, 2, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193, 15420, 17476, 17477, 32768, 65535, 65536, 65537]
while time.time() < t_end:
    n = rnd.choice(edges) if rnd.random() < 0.4 else rnd.randrange(1, min(N, 70000))
    n = min(n, N)
    off = rnd.randrange(0, N - n + 1)
    sc = scalars(n); w = wire(sc); want = expect(sc, off)
    out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
    if rnd.random() < 0.5:
        rc = lib.kzg_msm_g1_srs(ctx.handle, srs.handle, off, k._lib.ptr(w), n, k._lib.ptr(out), C.byref(inf))
        assert rc == 0, rc
    else:
        slot = rnd.randrange(0, 4)
        assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, off, k._lib.ptr(w), n, slot) == 0
        assert lib.kzg_msm_g1_srs_end(ctx.handle, slot, k._lib.ptr(out), C.byref(inf), None) == 0
    got = None if inf.value else pyref.point_from_wire(out)
    if got != want:
        print("MISMATCH seed", seed, "case", cases, "n", n, "off", off, flush=True); sys.exit(1)
    cases += 1
print("soak ok:", cases, "cases", flush=True)
