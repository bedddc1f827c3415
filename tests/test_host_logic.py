"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol that
include/kzg_bn254_mi355x.h declares, argument/error handling that needs no GPU, the host mirror's byte codecs,
and the host-side fold of partial sums (the multi-GPU epilogue).  No compute kernel is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as orc
import pyref
from pyref import R_

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def k():
    import rust_kzg_bn254_amd as k
    return k


def test_library_exports_every_declared_symbol(k):
    hdr = open(os.path.join(ROOT, "include", "kzg_bn254_mi355x.h")).read()
    declared = set(re.findall(r"\b(kzg_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"kzg_status"}
    assert len(declared) >= 20
    lib = C.CDLL(k._lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(k._lib.PROTOTYPES), declared ^ set(k._lib.PROTOTYPES)
    k.load()


def test_status_messages_are_the_reference_strings(k):
    lib = k.load()
    msg = lambda s: lib.kzg_status_message(s).decode()
    assert msg(k._lib.ERR_NOT_POWER_OF_TWO) == "length provided is not a power of 2"                 # kzg.rs:265-269
    assert msg(k._lib.ERR_DOMAIN) == "Could not perform IFFT due to domain consturction error"         # kzg.rs:276-278
    assert msg(k._lib.ERR_POLY_LENGTH) == "polynomial length is not correct"                          # kzg.rs:112-116
    assert msg(k._lib.ERR_ROOTS_LENGTH) == "inconsistent length between blob and root of unities"      # kzg.rs:135-139
    assert msg(k._lib.ERR_ZERO_LENGTH) == "Length of data after padding is 0"                          # helpers.rs:554-558
    assert msg(k._lib.ERR_TOO_LARGE) == "Input size exceeds maximum polynomial size"                   # polynomial.rs:42-46


def test_no_cpu_fallback_without_gpu(k):
    """On a box without a HIP device the product must fail loudly, never compute on the CPU."""
    lib = k.load()
    if lib.kzg_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert lib.kzg_ctx_create(0, C.byref(h)) == k._lib.ERR_NO_DEVICE and not h.value
    with pytest.raises(k.errors.DeviceError):
        k.Context(0)
    with pytest.raises(k.errors.DeviceError):
        k.KZG.new().calculate_and_store_roots_of_unity(64)


def test_null_and_argument_checks_need_no_gpu(k):
    lib = k.load()
    out = np.zeros(8, np.uint64)
    assert lib.kzg_msm_g1(None, None, 0, None, 0, k._lib.ptr(out), None) == k._lib.ERR_INVALID_ARG
    assert lib.kzg_fr_ntt(None, None, 4, 0) == k._lib.ERR_INVALID_ARG
    assert lib.kzg_g1_fold_partials(None, 0, None, None) == k._lib.ERR_INVALID_ARG


def test_fold_partials_host_epilogue(k, test_srs_points):
    """kzg_g1_fold_partials: XYZZ partials (as the GPUs return them) -> affine, incl. identity and doubling."""
    lib = k.load()
    pts = test_srs_points[:5]

    def xyzz(pt, zz_scale=1):
        if pt is None:
            return np.zeros(16, np.uint64)
        P = pyref.P
        lam = zz_scale % P                      # represent (x, y) as (x l^2, y l^3, l^2, l^3)
        return np.concatenate([pyref.fq_to_mont(pt[0] * lam * lam), pyref.fq_to_mont(pt[1] * lam ** 3),
                               pyref.fq_to_mont(lam * lam), pyref.fq_to_mont(lam ** 3)])

    parts = np.stack([xyzz(pts[0], 3), xyzz(pts[1], 12345), xyzz(None), xyzz(pts[0], 7), xyzz(pts[2])])
    out = np.zeros(8, np.uint64); inf = C.c_uint8(9)
    assert lib.kzg_g1_fold_partials(k._lib.ptr(parts), len(parts), k._lib.ptr(out), C.byref(inf)) == 0
    want = pyref.ec_add(pyref.ec_add(pyref.ec_mul(2, pts[0]), pts[1]), pts[2])
    assert pyref.point_from_wire(out) == want and inf.value == 0
    cancel = np.stack([xyzz(pts[3], 5), xyzz(pyref.ec_neg(pts[3]), 9)])
    assert lib.kzg_g1_fold_partials(k._lib.ptr(cancel), 2, k._lib.ptr(out), C.byref(inf)) == 0
    assert not out.any() and inf.value == 1
    assert lib.kzg_g1_fold_partials(None, 0, k._lib.ptr(out), C.byref(inf)) == 0 and inf.value == 1


def test_host_codecs_match_oracle_and_reference_vectors(k, kats, gettysburg):
    h = k.helpers
    for v in kats["pad_payload"]:
        raw = v["in"].encode()
        assert list(h.pad_payload(raw)) == v["out"]
        assert h.remove_internal_padding(h.pad_payload(raw))[:len(raw)] == raw
    assert h.pad_payload(gettysburg) == orc.pad_payload(gettysburg)
    assert len(h.remove_internal_padding(h.pad_payload(gettysburg))) == 1488        # helpers_test.rs:518-520
    with pytest.raises(k.errors.InvalidInputLength):
        h.remove_internal_padding(b"\x00" * 33)
    data = open(os.path.join(ROOT, "tests", "golden", "blobs.txt"), "rb").read()[:32 * 257]
    assert np.array_equal(h.to_fr_array(data), orc.to_fr_array(data))
    assert np.array_equal(h.to_fr_array(data[:40]), orc.to_fr_array(data[:40]))      # ragged tail
    els = h.to_fr_array(data[:320])
    assert h.to_byte_array(els, 320) == data[:320] and h.to_byte_array(els, 100) == data[:100]
    pt = pyref.point_to_wire((1, 2))
    assert h.serialize_compressed(pt) == orc.g1_serialize_compressed_ark(pt)
    assert h.serialize_compressed(np.zeros(8, np.uint64)) == orc.g1_serialize_compressed_ark(np.zeros(8, np.uint64))
    assert np.array_equal(h.hash_to_field_element(b"abc"), pyref.fr_to_mont(int.from_bytes(orc.sha256(b"abc"), "big")))
    for p in (0, 1, 20, 28):
        assert np.array_equal(h.get_primitive_root_of_unity(p), orc.fr_root_of_unity(p))
    with pytest.raises(k.errors.GenericError):
        h.get_primitive_root_of_unity(29)


def test_blob_and_polynomial_containers(k, gettysburg):
    """primitives/tests/blob_test.rs:21-53, polynomial_test.rs:67-84 (padding to the next power of two)."""
    blob = k.Blob.from_raw_data(gettysburg)
    assert len(blob) == 48 * 32 and blob.to_raw_data()[:len(gettysburg)] == gettysburg
    k.Blob(blob.data())                                        # canonical -> accepted
    with pytest.raises(k.errors.InvalidFieldElement):
        k.Blob(b"\xff" * 32)
    with pytest.raises(k.errors.InvalidInputLength):
        k.Blob(b"\x00" * 31)
    p = k.PolynomialEvalForm(pyref.frs_to_mont([1, 2, 3]))
    assert len(p) == 4 and p.len_underlying_blob_bytes() == 96 and pyref.frs_from_mont(p.evaluations()) == [1, 2, 3, 0]
    assert p.get_evalualtion(4) is None and not p.is_empty()
    c = k.PolynomialCoeffForm(pyref.frs_to_mont(list(range(1, 6))))
    assert len(c) == 8 and c.len_underlying_blob_field_elements() == 5
    assert c.to_bytes_be() == b"".join(v.to_bytes(32, "big") for v in range(1, 6))


def test_srs_order_guard(k):
    """prover/tests/kzg_test.rs:19-28."""
    with pytest.raises(k.errors.GenericError, match="Number of points to load exceeds SRS order."):
        k.SRS.new("tests/test-files/g1.point", 3000, 3001)
    with pytest.raises(k.errors.GenericError, match="Expected 3001 points"):
        k.SRS.new(os.path.join(ROOT, "tests", "golden", "g1.point"), 4000, 3001)


def test_bench_self_launch_takes_its_ranks_along_when_stopped():
    """`bench.py --gpus N` without a launcher is the parent of its ranks: when it is told to stop (the driver's time-out sends SIGTERM) no rank
    may stay behind.  CPU-only: the ranks are replaced by sleepers through the
    --test-child-cmd flag (an explicit argument only tests pass; round 4 read it from the environment: ADVICE r4)."""
    import signal
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KZG_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    p = subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--test-child-cmd", "sleep 300"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    deadline = time.time() + 60
    kids = []
    while time.time() < deadline and len(kids) < 2:
        time.sleep(0.3)
        out = subprocess.run(["ps", "-o", "pid=", "--ppid", str(p.pid)], capture_output=True, text=True).stdout.split()
        kids = [int(x) for x in out]
    assert len(kids) == 2, kids
    p.send_signal(signal.SIGTERM)
    rc = p.wait(timeout=30)
    assert rc == 128 + signal.SIGTERM
    time.sleep(0.5)
    for pid in kids:
        alive = subprocess.run(["ps", "-p", str(pid), "-o", "stat="], capture_output=True, text=True).stdout.strip()
        assert alive == "" or alive.startswith("Z"), (pid, alive)


def test_shard_steps_per_launch_by_world_size(k, monkeypatch):
    """How many steps of a commitment stream one launch carries (ShardedMsm.auto_group; host logic, the capacity rule is
    kzg_msm_batch_capacity): a 2^20-pair commitment over 1 / 2 / 4 / 8 ranks -> 1 / 2 / 2 / 4; every rank derives it from n // world, so
    uneven shards agree; the switches."""
    from rust_kzg_bn254_amd.sharding import ShardedMsm, shard_bounds
    monkeypatch.delenv("KZG_SHARD_GROUP", raising=False)
    monkeypatch.delenv("KZG_SHARD_GROUP_AUTO", raising=False)
    n = 1 << 20
    assert [ShardedMsm(None, n, 0, w).auto_group() for w in (1, 2, 4, 8)] == [1, 2, 2, 4]
    for world in (3, 5, 6, 7):
        groups = {ShardedMsm(None, n, r, world).auto_group() for r in range(world)}
        assert len(groups) == 1, (world, groups)
        assert sum(shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0] for r in range(world)) == n
    assert ShardedMsm(None, 1 << 12, 0, 1).auto_group() == 1               # below 2^13 pairs per rank: one launch per step
    monkeypatch.setenv("KZG_SHARD_GROUP_AUTO", "0")
    assert ShardedMsm(None, n, 0, 8).auto_group() == 1
    monkeypatch.setenv("KZG_SHARD_GROUP", "3")
    assert ShardedMsm(None, n, 0, 8).auto_group() == 3


def test_cpp_host_mirror_builds_and_fails_loudly_without_a_gpu(k, tmp_path):
    """include/kzg_bn254_mi355x.hpp (the C++ mirror of the reference's Rust API) compiles warning-free against the C header, and a program
    built on it has no CPU path to fall back to: without a HIP device its first call throws KzgError (device error)."""
    import subprocess
    exe = str(tmp_path / "reference_tests")
    libdir = os.path.join(ROOT, "rust-kzg-bn254_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "reference_tests.cpp"), "-L" + libdir, "-lkzg_bn254_mi355x", "-Wl,-rpath," + libdir, "-o", exe])
    if k._lib.load().kzg_device_count() > 0:
        pytest.skip("a GPU is present: tests/test_gpu_cpp_mirror.py runs the program")
    res = subprocess.run([exe, os.path.join(ROOT, "tests", "golden"), "%064x" % 255], capture_output=True, text=True, timeout=120)
    assert res.returncode == 3 and "no HIP device available (this library has no CPU fallback)" in res.stdout, (res.returncode, res.stdout, res.stderr)
