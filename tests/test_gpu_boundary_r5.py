"""-m gpu: round-5 additions to the boundary that are not part of the sharding work: the packed SRS file (kzg_srs_save_packed /
kzg_srs_load_packed -- SURVEY.md §5: the reference decodes its ceremony file at every SRS::new, prover/src/srs.rs:35-188)."""
import hashlib
import os

import numpy as np
import pytest

import oracle as orc
import pyref

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def k():
    import rust_kzg_bn254_amd as k
    k.load()
    k.default_context()
    return k


def test_packed_srs_round_trip_and_guards(k, tmp_path, test_srs_wire, gettysburg):
    """SRS::new from the reference's g1.point (GPU decompression) -> save_packed -> load_packed: the same 3000 points as
    srs.g1.points.string, the same commitment, a prefix load; every way a file can be wrong is refused with the right error."""
    from rust_kzg_bn254_amd.errors import DeserializationError, GenericError, NotOnCurveError
    srs = k.SRS.new(os.path.join(GOLDEN, "g1.point"), 3000, 3000)
    path = str(tmp_path / "srs.packed")
    srs.save_packed(path)
    raw = open(path, "rb").read()
    assert len(raw) == 56 + 3000 * 64 and raw[:8] == b"KZGSRS1\0" and int.from_bytes(raw[8:16], "little") == 3000
    assert raw[24:56] == hashlib.sha256(raw[56:]).digest()
    assert np.array_equal(np.frombuffer(raw[56:], dtype=np.uint64).reshape(-1, 8), test_srs_wire)      # the wire points themselves
    back = k.SRS.load_packed(path)
    assert len(back) == 3000 and np.array_equal(back.g1, test_srs_wire)
    blob = k.Blob.from_raw_data(gettysburg)
    kzg = k.KZG.new()
    c1, c2 = kzg.commit_blob(blob, srs), kzg.commit_blob(blob, back)
    rc, want = orc.commit_eval_form(test_srs_wire, blob.to_polynomial_eval_form().evaluations(), literal=False)
    assert np.array_equal(c1, c2) and np.array_equal(c1, want)
    part = k.SRS.load_packed(path, 1000)
    assert len(part) == 1000 and np.array_equal(part.g1, test_srs_wire[:1000])
    with pytest.raises(GenericError, match="exceeds SRS order"):
        k.SRS.load_packed(path, 3001)
    # an SRS with the identity in it survives the trip
    with_inf = np.ascontiguousarray(test_srs_wire[:5]).copy(); with_inf[2] = 0
    s_inf = k.SRS(with_inf); p_inf = str(tmp_path / "inf.packed"); s_inf.save_packed(p_inf)
    assert np.array_equal(k.SRS.load_packed(p_inf).g1, with_inf)
    empty = k.SRS(np.zeros((0, 8), np.uint64)); p_e = str(tmp_path / "empty.packed"); empty.save_packed(p_e)
    assert len(k.SRS.load_packed(p_e)) == 0

    def write(name, data):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        return p

    flipped = bytearray(raw); flipped[56 + 64 * 17 + 3] ^= 1                                   # one payload bit: digest mismatch
    for bad in (bytes(flipped), raw[:-1], raw + b"\0", b"NOTANSRS" + raw[8:], raw[:40]):
        with pytest.raises(DeserializationError):
            k.SRS.load_packed(write("bad.packed", bad))
    # a well-formed file (digest recomputed) whose point is not on the curve is caught on the device
    off = bytearray(raw); off[56 + 64 * 5] ^= 1
    off[24:56] = hashlib.sha256(bytes(off[56:])).digest()
    with pytest.raises(NotOnCurveError):
        k.SRS.load_packed(write("offcurve.packed", bytes(off)))
    with pytest.raises(GenericError, match="file could not be read or written"):
        k.SRS.load_packed(str(tmp_path / "does-not-exist.packed"))
    with pytest.raises(GenericError, match="file could not be read or written"):
        srs.save_packed(str(tmp_path / "no-such-dir" / "x.packed"))
    for s in (srs, back, part, s_inf, empty):
        s.close()
