"""CPU check of the PRODUCT's device math headers (rust-kzg-bn254_amd/csrc/field29.h, curve.h).

tests/hostcheck/hostcheck.cpp compiles those headers with g++ and -DKZG_BOUND_CHECK, which turns every
lazy-reduction bound the formulas rely on (limb magnitude and |a*b| < 2^261 m) into an abort().  The
results are compared bit for bit with the oracle.  This is a sanitizer-style build of the device math,
not a CPU fallback: nothing in the product loads libhostcheck.so.
"""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

import oracle as orc
import pyref
from pyref import P, R_

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "hostcheck", "hostcheck.cpp")
SO = os.path.join(HERE, "hostcheck", "libhostcheck.so")
CSRC = os.path.join(ROOT, "rust-kzg-bn254_amd", "csrc")

u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


@pytest.fixture(scope="module")
def hc():
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("field29.h", "curve.h", "field_constants.h", "naf.h", "fe_invert.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-DKZG_BOUND_CHECK", "-Wno-unknown-pragmas", "-fPIC",
                               "-shared", "-I" + CSRC, "-o", SO, SRC])
    return C.CDLL(SO)


def w32(limbs64):
    return np.ascontiguousarray(limbs64, dtype=np.uint64).view(np.uint32).copy()


def _mul(hc, which, a, b, square=False):
    a32, b32 = w32(a), w32(b)
    out = np.zeros(8, np.uint32)
    hc.hc_mul(which, a32.ctypes.data_as(u32p), b32.ctypes.data_as(u32p), out.ctypes.data_as(u32p), int(square))
    return out.view(np.uint64)


def test_field_constants_header_is_current():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_field_constants", os.path.join(ROOT, "tools", "gen_field_constants.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    assert mod.limbs29(P) == [0x187cfd47, 0x010460b6, 0x1c72a34f, 0x02d522d0, 0x1585d978, 0x02db40c0, 0x00a6e141, 0x0e5c2634, 0x0030644e]
    hdr = open(os.path.join(CSRC, "field_constants.h")).read()
    for m in (P, R_):
        inv = (-pow(m, -1, 1 << 29)) % (1 << 29)
        assert f"INV = 0x{inv:08x}u" in hdr
        for v in ((1 << 261) % m, (1 << 266) % m, (1 << 256) % m):
            assert ", ".join(f"0x{x:08x}u" for x in mod.limbs29(v)) in hdr


@pytest.mark.parametrize("which,mod", [(0, P), (1, R_)])
def test_mul_sqr_vs_oracle(hc, which, mod):
    rnd = random.Random(which)
    vals = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, (1 << 253) % mod, (1 << 29) - 1, 1 << 29, (1 << 232) - 1]
    vals += [rnd.randrange(mod) for _ in range(300)]
    for i in range(len(vals)):
        a = vals[i]; b = vals[(i * 7 + 3) % len(vals)]
        am, bm = pyref.to_limbs(a * (1 << 256) % mod), pyref.to_limbs(b * (1 << 256) % mod)
        got = _mul(hc, which, am, bm)
        assert np.array_equal(got, orc.f_mul(which, am, bm)), (a, b)
        assert np.array_equal(_mul(hc, which, am, am, square=True), orc.f_mul(which, am, am)), a


@pytest.mark.parametrize("which,mod", [(0, P), (1, R_)])
def test_lazy_signed_chain(hc, which, mod):
    rnd = random.Random(10 + which)
    cases = [(0, 0), (0, mod - 1), (mod - 1, 0), (mod - 1, mod - 1), (1, mod - 1)]
    cases += [(rnd.randrange(mod), rnd.randrange(mod)) for _ in range(200)]
    for a, b in cases:
        am, bm = w32(pyref.to_limbs(a * (1 << 256) % mod)), w32(pyref.to_limbs(b * (1 << 256) % mod))
        out = np.zeros(8, np.uint32)
        hc.hc_lazy(which, am.ctypes.data_as(u32p), bm.ctypes.data_as(u32p), out.ctypes.data_as(u32p))
        want = (2 * (a - b) - b) * (2 * a - b) % mod
        assert pyref.from_limbs(out.view(np.uint64)) == want * (1 << 256) % mod


def test_wire_to_canonical(hc):
    rnd = random.Random(5)
    for which, mod in ((0, P), (1, R_)):
        for a in [0, 1, mod - 1] + [rnd.randrange(mod) for _ in range(50)]:
            am = w32(pyref.to_limbs(a * (1 << 256) % mod)); out = np.zeros(8, np.uint32)
            hc.hc_wire_to_canonical(which, am.ctypes.data_as(u32p), out.ctypes.data_as(u32p))
            assert pyref.from_limbs(out.view(np.uint64)) == a


def _xyzz_to_affine(out32):
    """X, Y, ZZ, ZZZ wire -> affine python ints (None = identity)."""
    w = out32.view(np.uint64).reshape(4, 4)
    X, Y, ZZ, ZZZ = (pyref.fq_from_mont(w[i]) for i in range(4))
    if ZZ == 0:
        return None
    assert pow(ZZ, 3, P) == pow(ZZZ, 2, P)
    return (X * pow(ZZ, -1, P) % P, Y * pow(ZZZ, -1, P) % P)


def _chain(hc, pts, signs):
    wire = pyref.points_to_wire(pts).view(np.uint32).copy()
    sg = np.array(signs, dtype=np.uint8)
    out = np.zeros(32, np.uint32)
    hc.hc_madd_chain(wire.ctypes.data_as(u32p), sg.ctypes.data_as(u8p), C.c_size_t(len(pts)), out.ctypes.data_as(u32p))
    return _xyzz_to_affine(out)


def _ref_sum(pts, signs):
    acc = None
    for p, s in zip(pts, signs):
        acc = pyref.ec_add(acc, pyref.ec_neg(p) if s else p)
    return acc


def test_madd_chain_random_and_exceptional(hc, test_srs_points):
    rnd = random.Random(11)
    pts = test_srs_points[:200]
    signs = [rnd.randrange(2) for _ in pts]
    assert _chain(hc, pts, signs) == _ref_sum(pts, signs)
    # long chain keeps the stored-form bounds (bound check aborts otherwise)
    long_pts = [test_srs_points[rnd.randrange(3000)] for _ in range(3000)]
    long_s = [rnd.randrange(2) for _ in long_pts]
    assert _chain(hc, long_pts, long_s) == _ref_sum(long_pts, long_s)
    a, b = test_srs_points[5], test_srs_points[9]
    # P + P (doubling through the exceptional path), then more adds
    assert _chain(hc, [a, a], [0, 0]) == pyref.ec_mul(2, a)
    assert _chain(hc, [a, a, a, b], [0, 0, 0, 1]) == pyref.ec_add(pyref.ec_mul(3, a), pyref.ec_neg(b))
    assert _chain(hc, [a, b, pyref.ec_add(a, b)], [0, 0, 0]) == pyref.ec_mul(2, pyref.ec_add(a, b))
    # P + (-P) -> identity, then continue from the identity
    assert _chain(hc, [a, a], [0, 1]) is None
    assert _chain(hc, [a, a, b], [0, 1, 0]) == b
    assert _chain(hc, [a, pyref.ec_neg(a)], [0, 0]) is None
    # identity bases are skipped
    assert _chain(hc, [None, a, None], [0, 0, 1]) == a
    assert _chain(hc, [], []) is None
    # same point many times: 1P, 2P (dbl), 3P, ...
    assert _chain(hc, [a] * 17, [0] * 17) == pyref.ec_mul(17, a)
    assert _chain(hc, [a] * 17, [1] * 17) == pyref.ec_mul(R_ - 17, a)


def test_full_add_doubling_and_memory_format(hc, test_srs_points):
    rnd = random.Random(12)
    for trial in range(6):
        n = [2, 3, 10, 64, 65, 200][trial]
        pts = [test_srs_points[rnd.randrange(3000)] for _ in range(n)]
        signs = [rnd.randrange(2) for _ in pts]
        wire = pyref.points_to_wire(pts).view(np.uint32).copy(); sg = np.array(signs, dtype=np.uint8)
        for dbl in (0, 1, 5):
            out = np.zeros(32, np.uint32)
            hc.hc_add_halves(wire.ctypes.data_as(u32p), sg.ctypes.data_as(u8p), C.c_size_t(n), dbl, out.ctypes.data_as(u32p))
            assert _xyzz_to_affine(out) == pyref.ec_mul(1 << dbl, _ref_sum(pts, signs))
    # a + a through the full add (exceptional -> xyzz_dbl), a + (-a) -> identity
    a = test_srs_points[77]
    for pts, signs, want in (([a, a], [0, 0], pyref.ec_mul(2, a)), ([a, a], [0, 1], None)):
        wire = pyref.points_to_wire(pts).view(np.uint32).copy(); sg = np.array(signs, dtype=np.uint8)
        out = np.zeros(32, np.uint32)
        hc.hc_add_halves(wire.ctypes.data_as(u32p), sg.ctypes.data_as(u8p), C.c_size_t(2), 0, out.ctypes.data_as(u32p))
        assert _xyzz_to_affine(out) == want


def test_running_sum_shape(hc, test_srs_points):
    pts = test_srs_points[100:140] + [None, test_srs_points[3], None]
    wire = pyref.points_to_wire(pts).view(np.uint32).copy()
    out = np.zeros(32, np.uint32)
    hc.hc_running_sum(wire.ctypes.data_as(u32p), C.c_size_t(len(pts)), out.ctypes.data_as(u32p))
    want = None
    for k, p in enumerate(pts):
        want = pyref.ec_add(want, pyref.ec_mul(k + 1, p))
    assert _xyzz_to_affine(out) == want


def test_affine_wire_to_device_format(hc, test_srs_points):
    pt = test_srs_points[42]
    wire = pyref.point_to_wire(pt).view(np.uint32).copy(); out = np.zeros(16, np.uint32)
    hc.hc_affine_wire_to_device(wire.ctypes.data_as(u32p), out.ctypes.data_as(u32p))
    got = out.view(np.uint64).reshape(2, 4)
    assert pyref.from_limbs(got[0]) == pt[0] * (1 << 261) % P
    assert pyref.from_limbs(got[1]) == pt[1] * (1 << 261) % P
    zero = np.zeros(16, np.uint32); out[:] = 1
    hc.hc_affine_wire_to_device(zero.ctypes.data_as(u32p), out.ctypes.data_as(u32p))
    assert not out.any()


def test_paired_multiplies_equal_the_single_forms_on_extreme_limbs(hc):
    """k_msm_accumulate runs xyzz_madd on fe_mul2 / fe_sqr2; the host build now compiles the same paired form
    (curve.h no longer gates it on the device pass: test_madd_chain_* above run it under KZG_BOUND_CHECK), and here the paired
    routines are compared limb for limb with fe_mul / fe_sqr on the extreme SIGNED limb patterns the lazy formulas produce:
    all limbs +-(2^29 - 1), alternating signs, a (-6m, 9m) style difference, zero, one hot limb."""
    rnd = random.Random(29)
    top = (1 << 29) - 1
    hi = 1 << 22                                                  # top limb: keeps |value| < 2^254.1, i.e. |a * b| < 2^261 m
    pats = [[top] * 8 + [hi], [-top] * 8 + [-hi], [top if j % 2 else -top for j in range(8)] + [hi], [0] * 9, [1] + [0] * 8, [0] * 8 + [-(1 << 21)],
            [top] * 8 + [(1 << 22)], [-top] * 8 + [-(1 << 22)]]
    pats += [[rnd.randrange(-top, top + 1) for _ in range(8)] + [rnd.randrange(-(1 << 22), 1 << 22)] for _ in range(200)]
    arr = lambda v: (C.c_int32 * 9)(*v)                           # noqa: E731
    hc.hc_paired_vs_single.restype = C.c_int
    for i in range(len(pats)):
        a1, b1 = pats[i], pats[(3 * i + 1) % len(pats)]
        a2, b2 = pats[(5 * i + 2) % len(pats)], pats[(7 * i + 3) % len(pats)]
        # value bound of a product: |a * b| < 2^261 m; top limbs within +-2^22 keep |value| < 2^(232+22) * 1.01
        assert hc.hc_paired_vs_single(arr(a1), arr(b1), arr(a2), arr(b2)) == 0, i


def test_naf_recoding(hc):
    """naf.h: the digits of the MSM's NAF mode reproduce the scalar, are odd, at least w positions apart, below position 255."""
    rng = random.Random(2024)
    cap = 300
    pos = np.zeros(cap, np.uint32); key = np.zeros(cap, np.uint32); neg = np.zeros(cap, np.uint32)
    special = [0, 1, 2, 3, R_ - 1, R_ - 2, (1 << 253), (1 << 254) - 1, (1 << 254) - (1 << 200), 0xFFFFFFFF, (1 << 32), (1 << 64) - 1,
               int("01" * 127, 2), int("10" * 127, 2), int("0111" * 63, 2), (1 << 253) + (1 << 17) - 1, (1 << 248) - 1]
    for w in (10, 14, 16, 17, 18):
        assert hc.hc_naf_max_digits(w) == 254 // w + 1
        total = 0
        cases = special + [rng.randrange(R_) for _ in range(300)] + [rng.randrange(1 << rng.randrange(1, 254)) for _ in range(100)]
        for k in cases:
            words = np.array([(k >> (32 * j)) & 0xFFFFFFFF for j in range(8)], np.uint32)
            n = hc.hc_naf(words.ctypes.data_as(u32p), w, pos.ctypes.data_as(u32p), key.ctypes.data_as(u32p), neg.ctypes.data_as(u32p), cap)
            assert 0 <= n <= 254 // w + 1
            val, last = 0, None
            for t in range(n):
                p_, mag = int(pos[t]), 2 * int(key[t]) + 1
                assert mag < (1 << (w - 1)) and int(key[t]) < (1 << (w - 2)) and p_ <= 254
                assert last is None or p_ >= last + w
                last = p_
                val += (-mag if neg[t] else mag) << p_
            assert val == k, (w, hex(k))
            total += n
        if w == 18:
            assert total / len(cases) < 14.2          # ~254 / 19 on the random part


def test_reduce_small(hc):
    """fe_reduce_small (the NTT's product-free final reduction): +-2^k a for k <= 7 (|value| up to 128 (2m)) == big-integer value."""
    rng = random.Random(77)
    for which, m in ((0, P), (1, R_)):
        for a in [0, 1, m - 1, m // 2, (m - 1) // 3] + [rng.randrange(m) for _ in range(60)]:
            aw = w32(pyref.fq_to_mont(a) if which == 0 else pyref.frs_to_mont([a])[0])
            for k in (0, 1, 3, 6, 7):
                for neg in (0, 1):
                    out = np.zeros(8, np.uint32)
                    hc.hc_reduce_small(which, aw.ctypes.data_as(u32p), k, neg, out.ctypes.data_as(u32p))
                    want = ((-a if neg else a) << k) % m
                    ww = pyref.fq_to_mont(want) if which == 0 else pyref.frs_to_mont([want])[0]
                    assert np.array_equal(out.view(np.uint64), np.asarray(ww, dtype=np.uint64).reshape(-1)), (which, a, k, neg)


def test_safegcd_inverse(hc):
    """fe_inverse_safegcd (csrc/fe_invert.h, round 4: Bernstein-Yang division steps instead of a^(m-2) on the device) against big-integer
    inverses: edge values (0 -> 0, 1, m - 1, 2, (m +- 1) / 2, powers of two, values whose low limbs are all ones) and 3 000 random ones per
    field, canonical and lazy representatives; the bound-checked build asserts fe_mul's operand bounds at the exit products."""
    rng = random.Random(4242)
    for which, m in ((0, P), (1, R_)):
        edge = [0, 1, 2, 3, m - 1, m - 2, (m - 1) // 2, (m + 1) // 2, (1 << 29) - 1, 1 << 29, (1 << 58) - 1, (1 << 253) % m, (1 << 253) - 1,
                (1 << 261) % m, pow(5, (m - 1) // 3, m)]
        vals = edge + [rng.randrange(m) for _ in range(3000)] + [rng.randrange(1 << 64) for _ in range(200)]
        for a in vals:
            aw = w32(pyref.fq_to_mont(a) if which == 0 else pyref.frs_to_mont([a])[0])
            want = pow(a, -1, m) if a else 0
            ww = np.asarray(pyref.fq_to_mont(want) if which == 0 else pyref.frs_to_mont([want])[0], dtype=np.uint64).reshape(-1)
            for lazy in (0, 1):
                out = np.zeros(8, np.uint32)
                hc.hc_inverse_safegcd(which, aw.ctypes.data_as(u32p), out.ctypes.data_as(u32p), lazy)
                assert np.array_equal(out.view(np.uint64), ww), (which, a, lazy)
