"""CPU tests of the verifier row's host pairing (rust-kzg-bn254_amd/csrc/host_pairing.h) and of the host-only C-ABI
entry points kzg_pairings_verify / kzg_verify_proof (no GPU needed: O(1) host arithmetic inside the library).

Pins: (1) algebraic properties of the pairing (bilinearity, non-degeneracy, order r) from a g++ build of the header;
(2) end-to-end KZG relations on a known-tau setup whose commitments and proofs come from the oracle (CPU), following
verifier/tests/tests.rs:28-77.  `pairings_verify` only asks whether a product of pairings is the identity, a predicate
that every non-degenerate bilinear pairing on (G1, G2) decides identically, so these properties pin it completely."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle as orc
import pyref
from pyref import R_

import rust_kzg_bn254_amd as kzg
from rust_kzg_bn254_amd import _lib, helpers, verifier
from rust_kzg_bn254_amd.errors import GenericError, NotOnCurveError

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "rust-kzg-bn254_amd", "csrc")
GOLDEN = os.path.join(HERE, "golden")


def test_pairing_properties_host_build(tmp_path):
    exe = str(tmp_path / "pairingcheck")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + CSRC, os.path.join(HERE, "hostcheck", "pairingcheck.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split("\n")
    rows = dict(line.split() for line in out if line.strip())
    assert len(rows) == 37
    assert all(v == "1" for v in rows.values()), rows


def _g1(pt):
    return np.array(pyref.point_to_wire(pt), dtype=np.uint64).reshape(8)


TAU = 0x1F3A5C7E9B2D4F60718293A4B5C6D7E8F9


@pytest.fixture(scope="module")
def known_tau_srs():
    """64 powers [tau^i]G built by the oracle's scalar multiplication, and [tau]G2 from the library's G2 arithmetic.
    (The reference's verifier tests use the mainnet SRS file, which is not shipped in the reference tree here; the G1
    test fixture g1.point belongs to a different setup than consts::G2_TAU, so pairing tests need a self-made setup.)"""
    g = _g1((1, 2))
    pts = [np.asarray(orc.g1_scalar_mul(g, pyref.fr_to_mont(pow(TAU, i, R_))), dtype=np.uint64).reshape(8) for i in range(64)]
    return np.stack(pts), helpers.g2_mul_generator(kzg.fr.fr_from_int(TAU))


def test_srs_points_pair_with_g2_tau(known_tau_srs, test_srs_wire):
    """e(P_{i+1}, G2) == e(P_i, [tau]G2); and the mainnet G2_TAU does NOT pair with another setup's points."""
    srs, tau = known_tau_srs
    g2 = helpers.g2_generator()
    assert helpers.pairings_verify(srs[1], g2, srs[0], tau)
    assert helpers.pairings_verify(srs[7], g2, srs[6], tau)
    assert not helpers.pairings_verify(srs[2], g2, srs[0], tau)
    assert not helpers.pairings_verify(srs[1], tau, srs[0], g2)
    assert not helpers.pairings_verify(srs[1], g2, srs[0], helpers.g2_tau())
    other = np.asarray(test_srs_wire, dtype=np.uint64).reshape(-1, 8)
    assert not helpers.pairings_verify(other[1], g2, other[0], tau)


def _gettysburg_setup(srs, gettysburg):
    padded = orc.pad_payload(gettysburg)
    evals = orc.to_fr_array(padded)
    n = 64
    a = np.zeros((n, 4), dtype=np.uint64)
    a[: len(evals)] = np.asarray(evals, dtype=np.uint64).reshape(-1, 4)
    rc, roots = orc.calculate_roots_of_unity(len(padded))
    assert rc == n
    rc, commitment = orc.commit_eval_form(srs, a, literal=False)[:2]
    assert rc == 0
    return a, np.asarray(roots, dtype=np.uint64).reshape(-1, 4), np.asarray(commitment, dtype=np.uint64).reshape(8)


def test_oracle_proofs_verify(known_tau_srs, gettysburg):
    """verifier/tests/tests.rs:28-77 with CPU-side (oracle) commitment and proofs: verify_proof(commitment, proof_idx,
    evals[idx], roots[idx]) is true, and false for the next root / a wrong value; an off-domain point too."""
    srs, tau = known_tau_srs
    a, roots, commitment = _gettysburg_setup(srs, gettysburg)
    for idx in (0, 1, 17, 47):
        rc, proof, y = orc.compute_proof(srs, a, roots, roots[idx], literal=False)
        assert rc == 0
        proof = np.asarray(proof, dtype=np.uint64).reshape(8)
        assert verifier.verify_proof(commitment, proof, a[idx], roots[idx], tau) is True
        assert verifier.verify_proof(commitment, proof, a[idx], roots[(idx + 1) % 48], tau) is False
        assert verifier.verify_proof(commitment, proof, a[(idx + 1) % 48], roots[idx], tau) is False
    z = kzg.fr.fr_from_int(0xDEADBEEF12345)
    rc, proof, y = orc.compute_proof(srs, a, roots, z, literal=False)
    assert rc == 0
    assert verifier.verify_proof(commitment, np.asarray(proof, dtype=np.uint64).reshape(8), np.asarray(y, dtype=np.uint64).reshape(4), z, tau) is True
    assert verifier.verify_proof(commitment, np.asarray(proof, dtype=np.uint64).reshape(8), np.asarray(y, dtype=np.uint64).reshape(4), z) is False


def test_verify_proof_identity_points_and_invalid_points(test_srs_wire):
    """verifier/tests/tests.rs:383-409 (identity points are accepted as inputs), :412-457 (off-curve points rejected)."""
    srs = np.asarray(test_srs_wire, dtype=np.uint64).reshape(-1, 8)
    ident = np.zeros(8, dtype=np.uint64)
    one = kzg.fr.fr_from_int(1)
    two = kzg.fr.fr_from_int(2)
    assert verifier.verify_proof(ident, srs[3], one, two) in (True, False)
    assert verifier.verify_proof(srs[3], ident, one, two) in (True, False)
    # zero polynomial: C = identity, proof = identity, y = 0 holds at any z
    assert verifier.verify_proof(ident, ident, kzg.fr.fr_from_int(0), two) is True
    bad = srs[5].copy()
    bad[4] ^= 1
    with pytest.raises(NotOnCurveError, match="G1 point not on curve"):
        verifier.verify_proof(bad, srs[3], one, two)
    with pytest.raises(NotOnCurveError, match="G1 point not on curve"):
        verifier.verify_proof(srs[3], bad, one, two)


def test_verify_proof_custom_tau_and_tau_equals_z():
    """Known-tau setup: commitment to p(X) = 3 + 5X + 7X^2 built by oracle scalar muls; proof for z; tau == z is rejected
    (verify.rs:56-60)."""
    tau, z = 123456789123456789, 987654321
    coeffs = [3, 5, 7]
    p_tau = sum(c * pow(tau, i, R_) for i, c in enumerate(coeffs)) % R_
    y = sum(c * pow(z, i, R_) for i, c in enumerate(coeffs)) % R_
    q_tau = (p_tau - y) * pow(tau - z, -1, R_) % R_
    g = _g1((1, 2))
    commitment = np.asarray(orc.g1_scalar_mul(g, pyref.fr_to_mont(p_tau)), dtype=np.uint64).reshape(8)
    proof = np.asarray(orc.g1_scalar_mul(g, pyref.fr_to_mont(q_tau)), dtype=np.uint64).reshape(8)
    g2_tau = helpers.g2_mul_generator(kzg.fr.fr_from_int(tau))
    assert verifier.verify_proof(commitment, proof, kzg.fr.fr_from_int(y), kzg.fr.fr_from_int(z), g2_tau) is True
    assert verifier.verify_proof(commitment, proof, kzg.fr.fr_from_int(y + 1), kzg.fr.fr_from_int(z), g2_tau) is False
    assert verifier.verify_proof(commitment, proof, kzg.fr.fr_from_int(y), kzg.fr.fr_from_int(z), None) is False   # mainnet tau
    with pytest.raises(GenericError, match="Evaluation point equals trusted setup secret"):
        verifier.verify_proof(commitment, proof, kzg.fr.fr_from_int(y), kzg.fr.fr_from_int(tau), g2_tau)
    bad_tau = g2_tau.copy()
    bad_tau[0] ^= 1
    with pytest.raises(NotOnCurveError, match="G2_TAU not on curve"):
        verifier.verify_proof(commitment, proof, kzg.fr.fr_from_int(y), kzg.fr.fr_from_int(z), bad_tau)


def test_compute_r_powers_transcript_layout(test_srs_wire):
    """batch.rs:76-168: byte layout of the batch transcript, checked against an independent construction + the oracle's SHA-256."""
    srs = np.asarray(test_srs_wire, dtype=np.uint64).reshape(-1, 8)
    cs, ps = [srs[1], srs[2]], [srs[3], np.zeros(8, dtype=np.uint64)]
    zs = [kzg.fr.fr_from_int(11), kzg.fr.fr_from_int(R_ - 1)]
    ys = [kzg.fr.fr_from_int(5), kzg.fr.fr_from_int(0)]
    lens = [64, 4]
    got = verifier.compute_r_powers(cs, zs, ys, ps, lens)
    data = b"EIGENDA_RCKZGBATCH___V1_" + bytes(8) + (2).to_bytes(8, "big") + (64).to_bytes(8, "big") + (4).to_bytes(8, "big")
    for c, z, y, p in zip(cs, (11, R_ - 1), (5, 0), ps):
        data += orc.g1_serialize_compressed_ark(c) + z.to_bytes(32, "big") + y.to_bytes(32, "big") + orc.g1_serialize_compressed_ark(p)
    r = int.from_bytes(orc.sha256(data), "big") % R_
    assert kzg.fr.frs_to_ints(got) == [1, r]
    assert len(verifier.compute_r_powers([], [], [], [], [])) == 0


def test_g2_tau_matches_reference_g2_powers_fixture():
    """The reference tree ships mainnet G2 powers [tau^(2^i)]G2 (prover/tests/test-files/mainnet-data/g2.point.powerOf2, gnark
    compressed form: 64 bytes = X.A1 || X.A0 big-endian, top two bits = infinity / smaller-y / larger-y).  Entry 0 must be
    consts::G2_TAU (primitives/src/consts.rs:55-64) as this library holds it; every entry must lie on the twist."""
    P = pyref.P
    data = open(os.path.join(GOLDEN, "g2.point.powerOf2"), "rb").read()
    assert len(data) == 28 * 64

    def sqrt_fq(a):
        r = pow(a, (P + 1) // 4, P)
        return r if r * r % P == a % P else None

    def mul2(a, b):
        return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)

    def fq2_sqrt(a0, a1):
        if a1 == 0:
            r = sqrt_fq(a0)
            return (r, 0) if r is not None else (0, sqrt_fq((-a0) % P))
        alpha = sqrt_fq((a0 * a0 + a1 * a1) % P)
        assert alpha is not None
        half = pow(2, -1, P)
        x0 = sqrt_fq((a0 + alpha) * half % P)
        if x0 is None:
            x0 = sqrt_fq((a0 - alpha) * half % P)
        return (x0, a1 * pow(2 * x0, -1, P) % P)

    inv82 = pow(82, -1, P)
    b2 = (27 * inv82 % P, (-3 * inv82) % P)                       # 3 / (9 + u)
    larger = lambda v: (v[1] > (P - 1) // 2) if v[1] else (v[0] > (P - 1) // 2)     # noqa: E731
    points = []
    for i in range(28):
        ch = data[64 * i:64 * i + 64]
        flag = ch[0] >> 6
        assert flag in (2, 3)
        x = (int.from_bytes(ch[32:64], "big"), int.from_bytes(bytes([ch[0] & 0x3F]) + ch[1:32], "big"))
        x3 = mul2(mul2(x, x), x)
        rhs = ((x3[0] + b2[0]) % P, (x3[1] + b2[1]) % P)
        y = fq2_sqrt(*rhs)
        assert mul2(y, y) == rhs, i                               # on the twist
        neg = ((-y[0]) % P, (-y[1]) % P)
        y = (y if larger(y) else neg) if flag == 3 else (neg if larger(y) else y)
        points.append((x, y))
    lib_tau = helpers.g2_tau()
    got = tuple(kzg.fr.fq_to_int(lib_tau[4 * j:4 * j + 4]) for j in range(4))       # x.c0, x.c1, y.c0, y.c1
    assert got == (points[0][0][0], points[0][0][1], points[0][1][0], points[0][1][1])
    assert len(set(points)) == 28


def test_compute_r_powers_c_abi_matches_oracle_and_hashlib():
    """verifier/src/batch.rs:76-168 behind the C-ABI (`kzg_compute_r_powers`, host only) against the oracle's restatement and an
    independent hashlib transcript: identity points, n = 1 .. 300 (the row serialisation runs on a thread pool)."""
    import random
    import oracle as orc
    import rust_kzg_bn254_amd as k
    from rust_kzg_bn254_amd import verifier
    k.load()
    rnd = random.Random(0xB47C)
    base = [np.array(pyref.point_to_wire(pyref.ec_mul(rnd.randrange(1, pyref.R_), (1, 2))), dtype=np.uint64) for _ in range(12)]
    base.append(np.zeros(8, np.uint64))                      # the identity serialises as 0x40 in the last byte
    for n in (1, 2, 7, 300):
        cm = [base[rnd.randrange(len(base))] for _ in range(n)]
        pf = [base[rnd.randrange(len(base))] for _ in range(n)]
        zs = pyref.frs_to_mont([rnd.randrange(pyref.R_) for _ in range(n)])
        ys = pyref.frs_to_mont([rnd.choice([0, 1, pyref.R_ - 1, rnd.randrange(pyref.R_)]) for _ in range(n)])
        lens = [1 << rnd.randrange(0, 12) for _ in range(n)]
        got = verifier.compute_r_powers(cm, list(zs), list(ys), pf, lens)
        assert np.array_equal(got, orc.compute_r_powers(np.stack(cm), zs, ys, np.stack(pf), lens)), n
        assert np.array_equal(got, np.stack(verifier.compute_r_powers_py(cm, list(zs), list(ys), pf, lens))), n
        assert pyref.fr_from_mont(got[0]) == 1
        if n > 2:
            r = pyref.fr_from_mont(got[1])
            assert pyref.fr_from_mont(got[n - 1]) == pow(r, n - 1, pyref.R_)
    assert len(verifier.compute_r_powers([], [], [], [], [])) == 0


def test_pairing_known_answer_eip197_vector():
    """An EXTERNAL known answer for the pairing predicate (the reference holds none): the first vector of Ethereum's bn256 pairing
    precompile tests (EIP-197, go-ethereum `bn256Pairing` "jeff1": input of two (G1, G2) pairs whose pairing product is one), written
    down from public sources; every coordinate is checked to lie on its curve by big-integer arithmetic here, so a mistyped digit
    cannot pass.  e(P1, Q1) e(P2, Q2) == 1  <=>  pairings_verify(P1, Q1, -P2, Q2); the un-negated form and swapped G2 points must fail.
    EIP-197 encodes an Fq2 element imaginary part first; the library's wire format is c0 (real) | c1 (imaginary), Montgomery."""
    hexs = """1c76476f4def4bb94541d57ebba1193381ffa7aa76ada664dd31c16024c43f59 3034dd2920f673e204fee2811c678745fc819b55d3e9d294e45c9b03a76aef41
              209dd15ebff5d46c4bd888e51a93cf99a7329636c63514396b4a452003a35bf7 04bf11ca01483bfa8b34b43561848d28905960114c8ac04049af4b6315a41678
              2bb8324af6cfc93537a2ad1a445cfd0ca2a71acd7ac41fadbf933c2a51be344d 120a2a4cf30c1bf9845f20c6fe39e07ea2cce61f0c9bb048165fe5e4de877550
              111e129f1cf1097710d41c4ac70fcdfa5ba2023c6ff1cbeac322de49d1b6df7c 2032c61a830e3c17286de9462bf242fca2883585b93870a73853face6a6bf411
              198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2 1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed
              090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b 12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa""".split()
    v = [int(h, 16) for h in hexs]
    P = pyref.P
    assert all(c < P for c in v)
    for x, y in ((v[0], v[1]), (v[6], v[7])):                             # G1: y^2 = x^3 + 3
        assert (y * y - x ** 3 - 3) % P == 0

    def fq2_mul(a, b):                                                    # (a0 + a1 i)(b0 + b1 i), i^2 = -1
        return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)

    b_twist = fq2_mul((3, 0), (pow(9 * 9 + 1, -1, P) * 9 % P, (-pow(9 * 9 + 1, -1, P)) % P))     # 3 / (9 + i)
    for xi, xr, yi, yr in ((v[2], v[3], v[4], v[5]), (v[8], v[9], v[10], v[11])):                  # G2: y^2 = x^3 + 3 / (9 + i)
        x, y = (xr, xi), (yr, yi)
        x3 = fq2_mul(fq2_mul(x, x), x)
        y2 = fq2_mul(y, y)
        assert y2 == ((x3[0] + b_twist[0]) % P, (x3[1] + b_twist[1]) % P)

    def g1(x, y):
        return np.concatenate([pyref.fq_to_mont(x), pyref.fq_to_mont(y)]).astype(np.uint64)

    def g2(xi, xr, yi, yr):
        return np.concatenate([pyref.fq_to_mont(xr), pyref.fq_to_mont(xi), pyref.fq_to_mont(yr), pyref.fq_to_mont(yi)]).astype(np.uint64)

    p1, q1 = g1(v[0], v[1]), g2(v[2], v[3], v[4], v[5])
    p2, q2 = g1(v[6], v[7]), g2(v[8], v[9], v[10], v[11])
    p2neg = g1(v[6], (P - v[7]) % P)
    assert helpers.pairings_verify(p1, q1, p2neg, q2) is True
    assert helpers.pairings_verify(p1, q1, p2, q2) is False
    assert helpers.pairings_verify(p1, q2, p2neg, q1) is False
    # the second pair's G2 point is the generator: the constant the library serves
    assert np.array_equal(q2, helpers.g2_generator())
