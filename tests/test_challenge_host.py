"""Host-only (no GPU): `kzg_compute_challenge` -- the Fiat-Shamir transcript of helpers::compute_challenge
(primitives/src/helpers.rs:411-472) assembled and hashed in C -- against the oracle's transcript and an independent
hashlib construction; SHA-256 corner lengths through the same code."""
import ctypes as C
import hashlib
import random

import numpy as np
import pytest

import oracle as orc
import pyref
from pyref import P, R_

TAG = b"EIGENDA_FSBLOBVERIFY_V1_"


@pytest.fixture(scope="module")
def lib():
    import rust_kzg_bn254_amd as k
    return k._lib.load(), k


def challenge(lib_k, blob: bytes, commitment):
    lib, k = lib_k
    buf = np.frombuffer(blob, dtype=np.uint8).copy() if blob else np.zeros(1, np.uint8)
    z = np.zeros(4, np.uint64)
    c = np.ascontiguousarray(commitment, dtype=np.uint64).reshape(8)
    rc = lib.kzg_compute_challenge(buf.ctypes.data_as(k._lib.u8p), len(blob), k._lib.ptr(c), k._lib.ptr(z))
    return rc, z


def transcript_py(blob: bytes, pt):
    """tag || u64be(n) || n x 32 B (chunks mod r, big-endian) || ark-compressed commitment, by plain Python."""
    n_el = -(-len(blob) // 32)
    n = pyref.next_pow2(n_el)
    padded = blob + bytes(n_el * 32 - len(blob))
    evals = [int.from_bytes(padded[32 * i:32 * i + 32], "big") % R_ for i in range(n_el)] + [0] * (n - n_el)
    if pt is None:
        cb = bytes(31) + b"\x40"
    else:
        b = bytearray(pt[0].to_bytes(32, "little"))
        if pt[1] > (P - 1) // 2:
            b[31] |= 0x80
        cb = bytes(b)
    msg = TAG + n.to_bytes(8, "big") + b"".join(v.to_bytes(32, "big") for v in evals) + cb
    return int.from_bytes(hashlib.sha256(msg).digest(), "big") % R_


@pytest.mark.parametrize("length", [0, 1, 31, 32, 33, 64, 95, 1000, 32 * 64, 32 * 100 + 7, 32 * 4096])
def test_challenge_matches_python_and_oracle(lib, length):
    rnd = random.Random(length)
    blob = bytes(rnd.randrange(256) for _ in range(length))            # arbitrary bytes: most chunks are >= r and get reduced
    pt = pyref.ec_mul(rnd.randrange(1, R_), (1, 2))
    rc, z = challenge(lib, blob, pyref.point_to_wire(pt))
    assert rc == 0
    assert pyref.fr_from_mont(z) == transcript_py(blob, pt)
    if length:
        assert np.array_equal(z, orc.compute_challenge(blob, pyref.point_to_wire(pt)))


def test_challenge_canonical_blob_sign_flag_identity_and_errors(lib):
    _, k = lib
    rnd = random.Random(5)
    raw = bytes(rnd.randrange(32, 127) for _ in range(31 * 300 + 11))
    blob = orc.pad_payload(raw)                                        # canonical chunks: hashed straight from the buffer
    for s in (3, 4, 5, 6):                                             # both signs of y occur
        pt = pyref.ec_mul(s, (1, 2))
        rc, z = challenge(lib, blob, pyref.point_to_wire(pt))
        assert rc == 0 and pyref.fr_from_mont(z) == transcript_py(blob, pt)
        neg = (pt[0], P - pt[1])
        rc, z2 = challenge(lib, blob, pyref.point_to_wire(neg))
        assert rc == 0 and pyref.fr_from_mont(z2) == transcript_py(blob, neg) and not np.array_equal(z, z2)
    rc, z = challenge(lib, blob, np.zeros(8, np.uint64))               # identity commitment: is_on_curve() holds (helpers.rs:695)
    assert rc == 0 and pyref.fr_from_mont(z) == transcript_py(blob, None)
    bad = pyref.point_to_wire((1, 3))
    rc, _ = challenge(lib, blob, bad)
    assert rc == k._lib.ERR_G1_NOT_ON_CURVE
    # a chunk equal to r, r - 1, 2^256 - 1
    edge = R_.to_bytes(32, "big") + (R_ - 1).to_bytes(32, "big") + b"\xff" * 32 + bytes(32)
    pt = pyref.ec_mul(9, (1, 2))
    rc, z = challenge(lib, edge, pyref.point_to_wire(pt))
    assert rc == 0 and pyref.fr_from_mont(z) == transcript_py(edge, pt)


def test_transcript_prefix_two_streams_at_once(tmp_path):
    """csrc/host_transcript.h (round 6): the prefix tag || u64be(n) || n x 32 B as a segment generator, and SHA-256 over TWO prefixes interleaved in one
    thread (sha256rnds2 of two independent messages; what a pool job of the batch verifier does for a pair of blobs) -- every digest against hashlib over
    the prefix built in plain Python, for pairs of very different and of equal lengths, canonical and non-canonical chunks, ragged tails."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(os.path.dirname(here), "rust-kzg-bn254_amd", "csrc")
    so = str(tmp_path / "libtranscriptcheck.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + csrc, os.path.join(here, "hostcheck", "transcriptcheck.cpp"), "-o", so])
    tc = C.CDLL(so)
    u8p = C.POINTER(C.c_uint8)
    tc.tc_prefix_digests.argtypes = [u8p, C.c_size_t, C.c_size_t, u8p, C.c_size_t, C.c_size_t, u8p, u8p, u8p]

    def prefix_py(blob):
        n_el = -(-len(blob) // 32)
        n = pyref.next_pow2(n_el)
        padded = blob + bytes(n_el * 32 - len(blob))
        evals = [int.from_bytes(padded[32 * i:32 * i + 32], "big") % R_ for i in range(n_el)] + [0] * (n - n_el)
        return n, hashlib.sha256(TAG + n.to_bytes(8, "big") + b"".join(v.to_bytes(32, "big") for v in evals)).digest()

    rnd = random.Random(62)

    def blob_of(length, canonical):
        raw = bytearray(rnd.randrange(256) for _ in range(length))
        if canonical:
            for i in range(0, length, 32):
                raw[i] &= 0x1F
        elif length >= 64:
            raw[32:64] = (R_ + 5).to_bytes(32, "big")                       # a chunk just above r between two runs
        return bytes(raw)

    lengths = [1, 31, 32, 33, 64, 65, 96, 127, 1000, 4096, 32 * 100 + 7, 32 * 1024, 50000]
    for la in lengths:
        for lb in (lengths[(lengths.index(la) * 5 + 3) % len(lengths)], la, 32 * 2048 + 1):
            a, b = blob_of(la, rnd.random() < 0.5), blob_of(lb, rnd.random() < 0.5)
            (na, da), (nb_, db) = prefix_py(a), prefix_py(b)
            ba, bb = np.frombuffer(a, np.uint8).copy(), np.frombuffer(b, np.uint8).copy()
            o1, o2, o3 = (np.zeros(32, np.uint8) for _ in range(3))
            tc.tc_prefix_digests(ba.ctypes.data_as(u8p), la, na, bb.ctypes.data_as(u8p), lb, nb_, o1.ctypes.data_as(u8p), o2.ctypes.data_as(u8p), o3.ctypes.data_as(u8p))
            assert o1.tobytes() == da and o2.tobytes() == da and o3.tobytes() == db, (la, lb)

    # the run cap (512 chunks) and the top-word comparison of host_transcript.h: a chunk >= r AT, before and after every cap boundary of a long canonical run, and
    # chunks whose top eight bytes EQUAL r's (decided by the lower 24 bytes: r - 1 is canonical, r and r + 1 are reduced)
    base = bytearray(blob_of(32 * 1600 + 13, True))
    specials = {510: R_ + 7, 511: R_, 512: R_ - 1, 513: (1 << 256) - 1, 1023: R_ + 1, 1024: R_, 1025: R_ - 1, 1535: R_ - 1, 1536: R_ + (1 << 190), 1599: R_}
    for pos, v in specials.items():
        base[32 * pos:32 * pos + 32] = v.to_bytes(32, "big")
    variants = [bytes(base)]
    for pos in (511, 512, 1024):                                              # one special chunk alone in an otherwise canonical blob
        one = bytearray(blob_of(32 * 1600, True)); one[32 * pos:32 * pos + 32] = (R_ + 3).to_bytes(32, "big"); variants.append(bytes(one))
    for a in variants:
        b = blob_of(len(a) - 32 * 7, True)
        (na, da), (nb_, db) = prefix_py(a), prefix_py(b)
        ba, bb = np.frombuffer(a, np.uint8).copy(), np.frombuffer(b, np.uint8).copy()
        o1, o2, o3 = (np.zeros(32, np.uint8) for _ in range(3))
        tc.tc_prefix_digests(ba.ctypes.data_as(u8p), len(a), na, bb.ctypes.data_as(u8p), len(b), nb_, o1.ctypes.data_as(u8p), o2.ctypes.data_as(u8p), o3.ctypes.data_as(u8p))
        assert o1.tobytes() == da and o2.tobytes() == da and o3.tobytes() == db
