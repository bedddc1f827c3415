"""-m gpu: an SRS of 2^21 points, beyond the 24 index bits of the two-level sort: MSMs over it run as launches of 2^20 pairs with
compact indices (msm.hip: msm_launch_len / make_plan, msm_kernels.h: acc_point_index), several launches per call on one stream.
The reference takes SRS files of up to 2^28 points (prover/src/srs.rs:28-63).  Expected values: sum_i s_i tau^(offset+i) mod r by
big integers, times G1 (known-tau SRS) -- never another run of the HIP path."""
import ctypes as C
import hashlib
import random

import numpy as np
import pytest

import pyref
from pyref import R_

pytestmark = pytest.mark.gpu

TAU = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % R_
N = 1 << 21
MONT = (1 << 256) % R_


def to_wire(vals):
    return np.frombuffer(b"".join((v * MONT % R_).to_bytes(32, "little") for v in vals), dtype=np.uint64).reshape(-1, 4).copy()


def expect(scalars, offset=0):
    acc, tp = 0, pow(TAU, offset, R_)
    for s in scalars:
        acc = (acc + s * tp) % R_
        tp = tp * TAU % R_
    return pyref.ec_mul(acc, (1, 2))


@pytest.fixture(scope="module")
def env():
    import rust_kzg_bn254_amd as k
    k.load()
    ctx = k.default_context()
    srs = k.SRS.generate(TAU, N)
    yield k, ctx, srs
    srs.close()


def test_commitments_over_a_2_21_point_srs(env):
    k, ctx, srs = env
    lib = k._lib.load()
    rnd = random.Random(0xB16)
    out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
    # (offset, n): the whole SRS (two launches), a ragged length (2^20 + a short tail launch), one launch at an offset that straddles
    # the 2^20 boundary, a small MSM (second table set, compact indices) near the end of the SRS
    for offset, n in ((0, N), (0, (1 << 20) + 12_345), (900_000, 400_000), (N - 5000, 4096)):
        sc = [rnd.randrange(R_) for _ in range(n)]
        wire = to_wire(sc)
        want = expect(sc, offset)
        assert lib.kzg_msm_g1_srs(ctx.handle, srs.handle, offset, k._lib.ptr(wire), n, k._lib.ptr(out), C.byref(inf)) == 0
        assert pyref.point_from_wire(out) == want, ("sync", offset, n)
        # asynchronous form, two in flight on two slots
        out2 = np.zeros(8, np.uint64)
        assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, offset, k._lib.ptr(wire), n, 0) == 0
        assert lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, offset, k._lib.ptr(wire), n, 1) == 0
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 0, k._lib.ptr(out), C.byref(inf), None) == 0
        assert lib.kzg_msm_g1_srs_end(ctx.handle, 1, k._lib.ptr(out2), C.byref(inf), None) == 0
        assert pyref.point_from_wire(out) == want and pyref.point_from_wire(out2) == want, ("async", offset, n)


def test_skewed_scalars_over_the_2_21_point_srs(env):
    """few distinct scalar values: every coarse bin of the two-level sort is either empty or LARGE (tiled pass 2), heavy buckets"""
    k, ctx, srs = env
    lib = k._lib.load()
    n = (1 << 20) + 777
    vals = [R_ - 1, 1, (1 << 253) + 12345, 0]
    sc = [vals[(i * 7 + i // 3) % 4] for i in range(n)]
    out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
    wire = to_wire(sc)
    assert lib.kzg_msm_g1_srs(ctx.handle, srs.handle, 3, k._lib.ptr(wire), n, k._lib.ptr(out), C.byref(inf)) == 0
    assert pyref.point_from_wire(out) == expect(sc, 3)


@pytest.mark.parametrize("log_srs", [16, 18])
def test_two_level_sort_with_skewed_and_ragged_inputs(log_srs):
    """The two-level sort on its ordinary sizes (c = 15 tables at 2^16 points, c = 17 at 2^18): ragged lengths (tiles and chunks not
    full), all-equal scalars (15 buckets hold everything: every non-empty coarse bin is LARGE), a handful of values, zeros, r - 1."""
    import rust_kzg_bn254_amd as k
    k.load()
    ctx = k.default_context()
    lib = k._lib.load()
    n_srs = 1 << log_srs
    srs = k.SRS.generate(TAU, n_srs)
    try:
        rnd = random.Random(log_srs)
        out = np.zeros(8, np.uint64); inf = C.c_uint8(0)
        cases = []
        n = n_srs - 1234
        cases.append((0, [R_ - 1] * n))
        cases.append((1234, [rnd.randrange(R_)] * n))
        few = [rnd.randrange(R_) for _ in range(5)] + [0, 1]
        cases.append((7, [few[rnd.randrange(7)] for _ in range(n)]))
        cases.append((0, [rnd.randrange(R_) if i % 50 else 0 for i in range(n_srs)]))
        cases.append((100, [rnd.randrange(1 << 40) for _ in range(n_srs // 2 + 3)]))       # short scalars: the high windows are empty
        for offset, sc in cases:
            wire = to_wire(sc)
            assert lib.kzg_msm_g1_srs(ctx.handle, srs.handle, offset, k._lib.ptr(wire), len(sc), k._lib.ptr(out), C.byref(inf)) == 0
            want = expect(sc, offset)
            got = None if inf.value else pyref.point_from_wire(out)
            assert got == want, (log_srs, offset, len(sc))
    finally:
        srs.close()


def test_randomised_soak_of_the_srs_msm_entry_points():
    """tools/soak_msm.py for a few seconds with a fixed seed: random lengths (incl. the tile / chunk / table-set boundaries), offsets,
    scalar shapes and call forms against big-integer values (a 90-second run of it with random seeds is part of the round's checks)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SOAK_SECONDS="4", SOAK_SEED="20260102", SOAK_SRS_LOG="18")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_msm.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "soak ok" in r.stdout, r.stdout[-400:] + r.stderr[-400:]
