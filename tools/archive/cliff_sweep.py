"""Hunt for performance cliffs: latency of one MSM over (SRS size, MSM length, scalar shape); prints every case and flags the slow ones."""
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
def scal(kind, n):
    if kind == "uniform": return [rnd.randrange(R_) for _ in range(n)]
    if kind == "blob": return [int.from_bytes(bytes([0] + [rnd.randrange(32, 127) for _ in range(31)]), "big") for _ in range(n)]
    if kind == "same": return [rnd.randrange(R_)] * n
    if kind == "small": return [rnd.randrange(1 << 16) for _ in range(n)]
    if kind == "few": 
        f = [rnd.randrange(R_) for _ in range(4)]; return [rnd.choice(f) for _ in range(n)]
    if kind == "ones": return [1] * n
for srs_log in [int(x) for x in os.environ.get("SRS_LOGS", "9,11,13,15,17").split(",")]:
    N = 1 << srs_log
    srs = k.SRS.generate(TAU, N, ctx=ctx)
    for n in sorted({N, N // 2, max(1, N // 8), min(N, 100)}):
        row = []
        for kind in ("uniform", "blob", "same", "small", "few", "ones"):
            w = wire(scal(kind, n)); out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
            def one(): assert lib.kzg_msm_g1_srs(ctx.handle, srs.handle, 0, k._lib.ptr(w), n, k._lib.ptr(out), C.byref(inf)) == 0
            one(); one()
            t0 = time.perf_counter(); one(); one(); one(); dt = (time.perf_counter() - t0) / 3 * 1e3
            row.append("%s %.3f" % (kind, dt))
        ts = [float(x.split()[1]) for x in row]
        flag = "  <-- CLIFF" if max(ts) > 3 * min(ts) + 0.1 else ""
        print("SRS 2^%d n %6d: %s%s" % (srs_log, n, "  ".join(row), flag), flush=True)
    srs.close()
