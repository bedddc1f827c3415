"""Randomised soak of KZG::g1_ifft against the closed form on known-tau point sets: P_j = tau^(f + j) G  =>
L_i = tau^f (tau^n - 1) / n * w^i / (tau - w^i) G.  Random n = 2^1 .. 2^12, random first power f, SRS with and without per-bit tables (the
table paths need >= 2^15 points), the library's g1_ifft (its KZG_G1FFT_* A/B switches went in round 6).  SOAK_SECONDS (default 60)."""
import hashlib, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import pyref
from pyref import R_
import rust_kzg_bn254_amd as k
ctx = k.Context(0)
seed = int(os.environ.get("SOAK_SEED", str(int(time.time()))))
rnd = random.Random(seed)
print("seed", seed, flush=True)
G = (1, 2)
kzg = k.KZG.new(ctx)
t_end = time.time() + float(os.environ.get("SOAK_SECONDS", "60"))
SWITCHES = [{}]                                        # (round 6 removed the KZG_G1FFT_* switches with the rejected paths: the default path is what ships)
cases = 0
while time.time() < t_end:
    tau = rnd.randrange(2, R_)
    f = rnd.choice([0, 0, 1, rnd.randrange(1 << 20)])
    big = rnd.random() < 0.5
    size = (1 << 15) if big else 1 << rnd.randrange(1, 13)
    srs = k.SRS.generate(tau, size, ctx=ctx, first_power=f)
    try:
        for _ in range(6):
            log_n = rnd.randrange(1, min(12, size.bit_length() - 1) + 1)
            n = 1 << log_n
            env = rnd.choice(SWITCHES)
            for a, b in env.items(): os.environ[a] = b
            try:
                L = kzg.g1_ifft(n, srs)
            finally:
                for a in env: del os.environ[a]
            w = pyref.root_of_unity(log_n)
            pre = pow(tau, f, R_) * (pow(tau, n, R_) - 1) % R_ * pow(n, -1, R_) % R_
            idx = range(n) if n <= 16 else sorted({0, 1, n // 2, n - 1} | {rnd.randrange(n) for _ in range(6)})
            for i in idx:
                wi = pow(w, i, R_)
                want = pyref.ec_mul(pre * wi % R_ * pow(tau - wi, -1, R_) % R_, G)
                assert pyref.point_from_wire(L[i]) == want, (seed, tau, f, size, n, env, i)
            cases += 1
    finally:
        srs.close()
print("g1_ifft soak ok: %d cases" % cases, flush=True)
