"""Debug harness for the NAF mode of the table MSM: single special scalars, all-equal vectors, random vectors on a known-tau SRS."""
import ctypes as C, os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import pyref
from pyref import R_
import rust_kzg_bn254_amd as k
k.load(); k.default_context()
TAU = int.from_bytes(__import__("hashlib").sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % R_

def msm(srs, vals):
    sc = np.ascontiguousarray(pyref.frs_to_mont(vals), dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(8, np.uint64); inf = C.c_uint8(7)
    rc = k._lib.load().kzg_msm_g1_srs(srs.ctx.handle, srs.handle, 0, k._lib.ptr(sc), len(sc), k._lib.ptr(out), C.byref(inf))
    assert rc == 0, rc
    return pyref.point_from_wire(out)

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << log_n
srs = k.SRS.generate(TAU, n)
c = 12
half = sum(1 << (c * w + c - 1) for w in range(0, 254 // c + 1))
alt = sum(1 << (c * w + c - 1) for w in range(0, 254 // c + 1, 2))
full = (1 << 254) - 1
pats = [half % R_, (half - 1) % R_, full % R_, R_ - 1, R_ - 2, alt % R_, 1 << 253, (1 << 253) - 1, (1 << (c * 3)) - 1, 1, 0, 0xFFFFFFFF, 1 << 32, 3 << 31, 3 * (1 << 36) - 1, (1 << 253) - 3]
geo = (pow(TAU, n, R_) - 1) * pow(TAU - 1, -1, R_) % R_
tp = [1]
for i in range(1, n): tp.append(tp[-1] * TAU % R_)
def check(vals):
    ptau = sum(v * t for v, t in zip(vals, tp)) % R_
    return msm(srs, vals) == pyref.ec_mul(ptau, (1, 2))
import itertools
pairs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]] or [(7, 8)]
for a, b in pairs:
    vals = [pats[a] if i % 2 == 0 else pats[b] for i in range(n)]
    try:
        print("pair", a, b, check(vals), flush=True)
    except AssertionError as e:
        print("pair", a, b, "rc", e, flush=True)
