"""-m gpu: round-4 completions of the boundary.

* `kzg_verify_blob_kzg_proof` -- verify::verify_blob_kzg_proof (verifier/src/verify.rs:76-98) as ONE C call: against the composition
  of the reference's own steps (three calls), against the batch form at n = 1, and its error behaviour.
* `kzg_srs_load_compressed_ark_le` -- the `is_native = true` format of SRS::parallel_read_g1_points_native
  (prover/src/srs.rs:205-251, primitives/src/traits.rs:34-36): the reference's 3000 test points re-encoded in ark-serialize's compressed
  form must decode to srs.g1.points.string; malformed encodings are rejected."""
import os

import numpy as np
import pytest

import oracle as orc
import pyref
from pyref import R_

pytestmark = pytest.mark.gpu
TAU = int.from_bytes(__import__("hashlib").sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % R_
ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def k():
    import rust_kzg_bn254_amd as k
    k.load()
    k.default_context()
    return k


@pytest.fixture(scope="module")
def srs(k):
    return k.SRS.generate(TAU, 8192)


@pytest.fixture(scope="module")
def g2_tau(k):
    return k.helpers.g2_mul_generator(k.fr.fr_from_int(TAU))


def _prove(k, srs, raw):
    kz = k.KZG.new()
    blob = k.Blob.from_raw_data(raw)
    kz.calculate_and_store_roots_of_unity(len(blob))
    commitment = kz.commit_eval_form(blob.to_polynomial_eval_form(), srs)
    return blob, commitment, kz.compute_blob_proof(blob, commitment, srs)


def test_verify_blob_kzg_proof_one_call_matches_the_composition_and_the_batch(k, srs, g2_tau, gettysburg):
    rng = np.random.default_rng(11)
    raws = [gettysburg, b"x", bytes(rng.integers(32, 127, size=40000, dtype=np.uint8)),
            bytes(rng.integers(32, 127, size=31 * 8192 - 40, dtype=np.uint8))]          # the last: 8192 elements, beyond the batched evaluation kernel
    rows = [_prove(k, srs, raw) for raw in raws]
    for i, (b, c, p) in enumerate(rows):
        assert k.verify_blob_kzg_proof(b, c, p, g2_tau) is True, i
        assert k.verifier.verify_blob_kzg_proof_composed(b, c, p, g2_tau) is True, i
        assert k.verify_blob_kzg_proof_batch([b], [c], [p], g2_tau) is True, i
        other = rows[(i + 1) % len(rows)]
        for bb, cc, pp in ((b, c, other[2]), (b, other[1], p), (other[0], c, p)):        # wrong proof / commitment / blob
            one = k.verify_blob_kzg_proof(bb, cc, pp, g2_tau)
            assert one is False and one == k.verifier.verify_blob_kzg_proof_composed(bb, cc, pp, g2_tau) == k.verify_blob_kzg_proof_batch([bb], [cc], [pp], g2_tau)
        if not k.fr.g1_is_identity(p):                                                   # (a constant polynomial's proof is the identity: e(C - yG, G2) = 1 under ANY setup)
            assert k.verify_blob_kzg_proof(b, c, p) is False                             # consts::G2_TAU is another setup


def test_verify_blob_kzg_proof_errors(k, srs, g2_tau, gettysburg):
    from rust_kzg_bn254_amd.errors import GenericError, NotOnCurveError
    b, c, p = _prove(k, srs, gettysburg)
    off = np.array(pyref.point_to_wire((1, 3)), dtype=np.uint64)
    with pytest.raises(NotOnCurveError, match="G1 point not on curve"):                  # verify.rs:82 -> helpers.rs:694-699
        k.verify_blob_kzg_proof(b, off, p, g2_tau)
    with pytest.raises(NotOnCurveError, match="G1 point not on curve"):                  # verify.rs:85
        k.verify_blob_kzg_proof(b, c, off, g2_tau)
    empty = k.Blob(b"")
    with pytest.raises(GenericError):                                                    # helpers.rs:554-558 through evaluate_polynomial_in_evaluation_form
        k.verify_blob_kzg_proof(empty, c, p, g2_tau)
    bad_tau = np.zeros(16, np.uint64); bad_tau[0] = 5
    with pytest.raises(NotOnCurveError, match="G2_TAU not on curve"):                    # verify.rs:29-33
        k.verify_blob_kzg_proof(b, c, p, bad_tau)
    ident = np.zeros(8, np.uint64)
    assert k.verify_blob_kzg_proof(b, c, ident, g2_tau) is False                         # identity proof: a bool, not an error (tests.rs:272-311)


def ark_compress(pt):
    """ark-serialize 0.5 compressed G1Affine: x little-endian, 0x80 = y > -y, 0x40 = infinity (restated)."""
    if pt is None:
        return bytes(31) + bytes([0x40])
    x, y = pt
    b = bytearray(x.to_bytes(32, "little"))
    if y > pyref.P - y:
        b[31] |= 0x80
    return bytes(b)


def test_srs_load_ark_le_matches_the_reference_points(k, tmp_path, test_srs_points, test_srs_wire):
    pts = list(test_srs_points)
    pts[7] = None                                                                         # an encoded identity among them
    path = tmp_path / "g1.ark"
    path.write_bytes(b"".join(ark_compress(p) for p in pts))
    srs = k.SRS.new(str(path), 3000, 3000, is_native=True)
    want = test_srs_wire.copy(); want[7] = 0
    assert np.array_equal(srs.g1, want)
    # the same bytes as the oracle's serialiser writes them (the convention the Fiat-Shamir transcripts use)
    for i in (0, 1, 2999):
        assert orc.g1_serialize_compressed_ark(test_srs_wire[i]) == ark_compress(test_srs_points[i])
    # and through the MSM path: a commitment over the loaded SRS equals the oracle's over the decimal points
    sc = pyref.frs_to_mont([(i * 0x9E3779B97F4A7C15 + 12345) % R_ for i in range(512)])
    kz = k.KZG.new()
    pts8 = test_srs_wire[:512].copy(); pts8[7] = 0
    got = kz.commit_coeff_form(k.PolynomialCoeffForm(sc), srs)
    assert np.array_equal(got, orc.msm_pippenger(pts8, sc))
    srs.close()


def test_srs_load_ark_le_rejects_malformed_points(k, tmp_path, test_srs_points):
    from rust_kzg_bn254_amd.errors import DeserializationError
    good = [ark_compress(p) for p in test_srs_points[:8]]

    def load(rows):
        path = tmp_path / "bad.ark"
        path.write_bytes(b"".join(rows))
        return k.SRS.new(str(path), len(rows), len(rows), is_native=True)

    both = bytearray(good[3]); both[31] |= 0xC0                                           # both flags: SWFlags::from_u8 -> None
    with pytest.raises(DeserializationError):
        load(good[:3] + [bytes(both)] + good[4:])
    big = bytearray(pyref.P.to_bytes(32, "little"))                                       # x = p: not a canonical field element
    with pytest.raises(DeserializationError):
        load(good[:5] + [bytes(big)])
    x = 1
    while pow((x ** 3 + 3) % pyref.P, (pyref.P - 1) // 2, pyref.P) == 1:                  # an x with no point on the curve
        x += 1
    with pytest.raises(DeserializationError, match="Deserialization failed"):          # deserialize_compressed reports every bad encoding alike: traits.rs:34-36
        load([x.to_bytes(32, "little")] + good[:2])
    flipped = bytearray(good[2]); flipped[31] ^= 0x80                                     # the other root: a valid point, -P
    s = load([bytes(flipped)])
    xw, yw = pyref.point_from_wire(s.g1[0])
    assert (xw, yw) == (test_srs_points[2][0], pyref.P - test_srs_points[2][1])
    s.close()


def test_device_caches_are_rebuilt_after_the_last_context_is_destroyed(k):
    """The twiddle tables of the NTT and the scalar / digit lists of g1_ifft are process-wide caches keyed by device; they are released
    when the LAST context of the device is destroyed and rebuilt by the next one: same results before and after.  Runs in a
    child process so that no other context of this test session keeps the caches alive."""
    import subprocess
    import sys
    code = (
        "import sys, hashlib\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, rust_kzg_bn254_amd as k\n"
        "from rust_kzg_bn254_amd import _lib\n"
        "lib = _lib.load()\n"
        "def run():\n"
        "    ctx = k.Context(0)\n"
        "    srs = k.SRS.generate(%d, 1 << 12, ctx=ctx)\n"
        "    a = np.arange(4 << 14, dtype=np.uint64).reshape(-1, 4) %% 251\n"
        "    assert lib.kzg_fr_ntt(ctx.handle, _lib.ptr(a), 1 << 14, 0) == 0\n"
        "    lag = np.zeros((1024, 8), np.uint64)\n"
        "    assert lib.kzg_g1_ifft(ctx.handle, srs.handle, 1024, _lib.ptr(lag)) == 0\n"
        "    d = hashlib.sha256(a.tobytes() + lag.tobytes()).hexdigest()\n"
        "    srs.close(); lib.kzg_ctx_destroy(ctx.handle); ctx.handle = None\n"
        "    return d\n"
        "first = run(); second = run(); third = run()\n"
        "assert first == second == third, (first, second, third)\n"
        "print('caches ok', first[:16])\n"
    ) % (ROOT_DIR, os.path.join(ROOT_DIR, "tests"), TAU)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout[-500:], res.stderr[-2000:])
    assert "caches ok" in res.stdout
