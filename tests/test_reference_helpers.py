"""primitives/tests/helpers_test.rs: the tests that tests/test_gpu_reference_tests.py does not hold, one for one and under their own names,
through the Python mirror of `primitives::helpers` (rust_kzg_bn254_amd/helpers.py).  Host-only functions (byte codecs, the G2 checks of the
library's host pairing code) run in the CPU suite; everything that decodes points, evaluates or hashes through the C-ABI with a context is
marked `gpu`.  Random G1 / G2 points are [s]G for seeded scalars (ark_std::test_rng() has no counterpart here); the assertions are the
reference's.  Also here: the reference functions that have no test of their own in the reference tree (is_zeroed is tested there; set_bytes_canonical,
str_vec_to_fr_vec, read_g1_point_from_bytes_be, SRS::process_chunks, KZG::compute_quotient_eval_on_domain are not) against pyref / the oracle's vectors."""
import os
import random

import numpy as np
import pytest

import pyref
from pyref import P as P_, R_

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GETTYSBURG = open(os.path.join(GOLDEN, "gettysburg.txt"), "rb").read()


@pytest.fixture(scope="module")
def k():
    import rust_kzg_bn254_amd as k
    k.load()
    return k


def rand_g2(k, rng):
    return k.helpers.g2_mul_generator(pyref.fr_to_mont(rng.randrange(2, R_)))


def rand_g1(rng):
    return pyref.point_to_wire(pyref.ec_mul(rng.randrange(2, R_), (1, 2)))


def fq2_sqrt(a0, a1):
    """sqrt of a0 + a1 u in Fq[u] / (u^2 + 1), p = 3 mod 4; None when there is none"""
    if a1 == 0:
        s = pow(a0, (P_ + 1) // 4, P_)
        if s * s % P_ == a0:
            return s, 0
        s = pow(-a0 % P_, (P_ + 1) // 4, P_)
        return (0, s) if s * s % P_ == -a0 % P_ else None
    norm = (a0 * a0 + a1 * a1) % P_
    s = pow(norm, (P_ + 1) // 4, P_)
    if s * s % P_ != norm:
        return None
    inv2 = pow(2, P_ - 2, P_)
    for cand in ((a0 + s) * inv2 % P_, (a0 - s) * inv2 % P_):
        x0 = pow(cand, (P_ + 1) // 4, P_)
        if x0 and x0 * x0 % P_ == cand:
            return x0, a1 * pow(2 * x0, P_ - 2, P_) % P_
    return None


def g2_wire(x0, x1, y0, y1):
    return np.concatenate([pyref.fq_to_mont(v) for v in (x0, x1, y0, y1)])


# ---- host-only ---------------------------------------------------------------------------------------------------------------------------
def test_g2_is_on_curve(k):                                                   # helpers_test.rs:375-387 (1000 points there, 200 here)
    rng = random.Random(375)
    for _ in range(200):
        point = rand_g2(k, rng)
        assert k.helpers.is_on_curve_g2(point)
        not_on_curve = point.copy()
        not_on_curve[0:4] = pyref.fq_to_mont((pyref.fq_from_mont(point[0:4]) + 1) % P_)        # x += Fq2::one()
        assert not k.helpers.is_on_curve_g2(not_on_curve)


def test_get_num_element(k):                                                  # helpers_test.rs:462-465
    assert k.helpers.get_num_element(1000, k.consts.BYTES_PER_FIELD_ELEMENT) == 32


def test_pad_payload(k):                                                      # helpers_test.rs:468-521
    padded = k.helpers.pad_payload(b"hi")
    assert list(padded) == [0, 104, 105] + [0] * 29
    assert list(k.helpers.remove_internal_padding(padded)) == [104, 105] + [0] * 29
    larger = k.helpers.pad_payload(b"zxcvbnm,.//asdfgghjkl;'][poiuytrewq`1234567890zxcvbnm,.//1234567890")
    assert list(larger) == [0, 122, 120, 99, 118, 98, 110, 109, 44, 46, 47, 47, 97, 115, 100, 102, 103, 103, 104, 106, 107, 108, 59, 39, 93, 91,
                            112, 111, 105, 117, 121, 116, 0, 114, 101, 119, 113, 96, 49, 50, 51, 52, 53, 54, 55, 56, 57, 48, 122, 120, 99, 118, 98,
                            110, 109, 44, 46, 47, 47, 49, 50, 51, 52, 53, 0, 54, 55, 56, 57, 48] + [0] * 26
    assert list(k.helpers.remove_internal_padding(larger)) == [122, 120, 99, 118, 98, 110, 109, 44, 46, 47, 47, 97, 115, 100, 102, 103, 103, 104,
                                                               106, 107, 108, 59, 39, 93, 91, 112, 111, 105, 117, 121, 116, 114, 101, 119, 113, 96,
                                                               49, 50, 51, 52, 53, 54, 55, 56, 57, 48, 122, 120, 99, 118, 98, 110, 109, 44, 46, 47,
                                                               47, 49, 50, 51, 52, 53, 54, 55, 56, 57, 48] + [0] * 26
    unpadded = k.helpers.remove_internal_padding(k.helpers.pad_payload(GETTYSBURG))
    assert len(unpadded) == 1488 and len(GETTYSBURG) <= len(unpadded)


def test_is_zeroed_all_zeroes(k):                                             # helpers_test.rs:524-529
    assert k.helpers.is_zeroed(0, [0, 0, 0, 0, 0])


def test_is_zeroed_first_byte_non_zero(k):                                    # :532-540
    assert not k.helpers.is_zeroed(1, [0, 0, 0, 0, 0])


def test_is_zeroed_buffer_non_zero(k):                                        # :543-551
    assert not k.helpers.is_zeroed(0, [0, 0, 1, 0, 0])


def test_is_zeroed_first_byte_and_buffer_non_zero(k):                         # :554-562
    assert not k.helpers.is_zeroed(1, [0, 1, 0, 0, 0])


def test_is_zeroed_empty_buffer(k):                                           # :565-573
    assert k.helpers.is_zeroed(0, [])


def test_is_zeroed_empty_buffer_non_zero_first_byte(k):                       # :576-584
    assert not k.helpers.is_zeroed(1, [])


def test_how_to_read_bytes(k):                                                # :452-459 (prints the limbs there; the value is checked here)
    the_bytes = bytes([31, 94, 220, 111, 30, 251, 22, 93, 69, 166, 84, 121, 141, 75, 170, 165, 14, 59, 77, 36, 24, 41, 19, 174, 245, 17, 10, 21, 88,
                       14, 186, 173])
    assert pyref.fr_from_mont(k.helpers.set_bytes_canonical(the_bytes)) == int.from_bytes(the_bytes, "big") % R_


def test_primitive_roots_from_bigint_to_fr(k):                                # :587-628 (the 29 decimal strings of the reference's test)
    data = ["1",
            "21888242871839275222246405745257275088548364400416034343698204186575808495616",
            "21888242871839275217838484774961031246007050428528088939761107053157389710902",
            "19540430494807482326159819597004422086093766032135589407132600596362845576832",
            "14940766826517323942636479241147756311199852622225275649687664389641784935947",
            "4419234939496763621076330863786513495701855246241724391626358375488475697872",
            "9088801421649573101014283686030284801466796108869023335878462724291607593530",
            "10359452186428527605436343203440067497552205259388878191021578220384701716497",
            "3478517300119284901893091970156912948790432420133812234316178878452092729974",
            "6837567842312086091520287814181175430087169027974246751610506942214842701774",
            "3161067157621608152362653341354432744960400845131437947728257924963983317266",
            "1120550406532664055539694724667294622065367841900378087843176726913374367458",
            "4158865282786404163413953114870269622875596290766033564087307867933865333818",
            "197302210312744933010843010704445784068657690384188106020011018676818793232",
            "20619701001583904760601357484951574588621083236087856586626117568842480512645",
            "20402931748843538985151001264530049874871572933694634836567070693966133783803",
            "421743594562400382753388642386256516545992082196004333756405989743524594615",
            "12650941915662020058015862023665998998969191525479888727406889100124684769509",
            "11699596668367776675346610687704220591435078791727316319397053191800576917728",
            "15549849457946371566896172786938980432421851627449396898353380550861104573629",
            "17220337697351015657950521176323262483320249231368149235373741788599650842711",
            "13536764371732269273912573961853310557438878140379554347802702086337840854307",
            "12143866164239048021030917283424216263377309185099704096317235600302831912062",
            "934650972362265999028062457054462628285482693704334323590406443310927365533",
            "5709868443893258075976348696661355716898495876243883251619397131511003808859",
            "19200870435978225707111062059747084165650991997241425080699860725083300967194",
            "7419588552507395652481651088034484897579724952953562618697845598160172257810",
            "2082940218526944230311718225077035922214683169814847712455127909555749686340",
            "19103219067921713944291392827692070036145651957329286315305642004821462161904"]
    fr_s = k.helpers.str_vec_to_fr_vec(data)
    for i in range(29):
        assert np.array_equal(k.helpers.get_primitive_root_of_unity(i), fr_s[i])
    with pytest.raises(k.errors.GenericError):
        k.helpers.get_primitive_root_of_unity(29)


def test_validate_g2_point_valid_point(k):                                    # :696-708
    rng = random.Random(696)
    for _ in range(10):
        k.helpers.example_validate_g2_point(rand_g2(k, rng))


def test_validate_g2_point_identity_point(k):                                 # :711-717
    with pytest.raises(k.errors.KzgError):
        k.helpers.example_validate_g2_point(np.zeros(16, dtype=np.uint64))


def test_validate_g2_point_invalid_curve_point(k):                            # :720-751
    invalid_point = g2_wire(1, 0, 1, 0)
    assert not k.helpers.is_on_curve_g2(invalid_point)
    with pytest.raises(k.errors.NotOnCurveError) as e:
        k.helpers.example_validate_g2_point(invalid_point)
    assert e.value.message == "G2 point not on curve"


def test_validate_point_functions_consistency(k):                             # :754-787
    rng = random.Random(754)
    for _ in range(5):
        g1_point = rand_g1(rng)
        manual = (not k.fr.g1_is_identity(g1_point)) and k.helpers.is_on_curve_g1(g1_point)      # cofactor 1: on the curve = in the subgroup
        try:
            k.helpers.validate_g1_point(g1_point)
            function_check = True
        except k.errors.KzgError:
            function_check = False
        assert manual == function_check
    gen = k.helpers.g2_generator()
    for _ in range(5):
        g2_point = rand_g2(k, rng)
        manual = g2_point.any() and k.helpers.is_on_curve_g2(g2_point) and not np.array_equal(g2_point, gen)
        try:
            k.helpers.example_validate_g2_point(g2_point)
            function_check = True
        except k.errors.KzgError:
            function_check = False
        assert manual == function_check


def test_validate_point_functions_generator_rejection(k):                     # :790-844
    g1_generator = pyref.point_to_wire((1, 2))
    assert not k.fr.g1_is_identity(g1_generator) and k.helpers.is_on_curve_g1(g1_generator)
    k.helpers.validate_g1_point(g1_generator)                                 # "G1 generator should not be rejected"
    g2_generator = k.helpers.g2_generator()
    assert g2_generator.any() and k.helpers.is_on_curve_g2(g2_generator)
    with pytest.raises(k.errors.G2GeneratorNotAcceptedError) as e:
        k.helpers.example_validate_g2_point(g2_generator)
    assert e.value.message == "G2 point cannot be the generator point"
    assert str(e.value) == "g2 generator not accepted error: G2 point cannot be the generator point"          # errors.rs:56-57


def test_validate_g2_point_outside_the_subgroup(k):
    """helpers.rs:753-757 (no test in the reference reaches it): a point of the twist outside the order-r subgroup (cofactor 2p - r: almost every
    curve point) -> NotOnCurveError("G2 point not in correct subgroup")"""
    # b' = 3 / (9 + u)
    d = pow(82, P_ - 2, P_)
    b0, b1 = 27 * d % P_, -3 * d % P_
    rng = random.Random(753)
    found = 0
    while found < 3:
        x0, x1 = rng.randrange(P_), rng.randrange(P_)
        xx0, xx1 = (x0 * x0 - x1 * x1) % P_, 2 * x0 * x1 % P_
        c0, c1 = (xx0 * x0 - xx1 * x1 + b0) % P_, (xx0 * x1 + xx1 * x0 + b1) % P_
        y = fq2_sqrt(c0, c1)
        if y is None:
            continue
        pt = g2_wire(x0, x1, y[0], y[1])
        assert k.helpers.is_on_curve_g2(pt)
        with pytest.raises(k.errors.NotOnCurveError) as e:
            k.helpers.example_validate_g2_point(pt)
        assert e.value.message == "G2 point not in correct subgroup"
        found += 1


def test_str_vec_to_fr_vec(k):                                                # helpers.rs:134-149
    out = k.helpers.str_vec_to_fr_vec(["-1", "0", "1", str(R_ + 5), "12345678901234567890123456789"])
    assert pyref.frs_from_mont(out) == [R_ - 1, 0, 1, 5, 12345678901234567890123456789]
    for bad in ("", "12a", "0x10", " 1"):
        with pytest.raises(ValueError, match="could not load string to Fr"):
            k.helpers.str_vec_to_fr_vec([bad])


def test_set_bytes_canonical(k):                                              # helpers.rs:32-34 (Fr::from_be_bytes_mod_order: any length)
    for data in (b"", b"\x01", bytes(range(1, 32)), b"\xff" * 32, b"\xff" * 40):
        assert pyref.fr_from_mont(k.helpers.set_bytes_canonical(data)) == int.from_bytes(data, "big") % R_


def test_validate_blob_data_vectorised_positions(k):                          # helpers.rs:784-810: the position in the message is the first bad chunk
    good = (R_ - 1).to_bytes(32, "big")
    for bad_value in (R_, R_ + 1, (1 << 256) - 1, R_ + (1 << 64), R_ + (1 << 128), R_ + (1 << 192)):
        data = good * 3 + bad_value.to_bytes(32, "big") + good + bad_value.to_bytes(32, "big")
        with pytest.raises(k.errors.InvalidFieldElement) as e:
            k.helpers.validate_blob_data_as_canonical_field_elements(data)
        assert e.value.message == "Field element at position 3 is not canonical or invalid"
    k.helpers.validate_blob_data_as_canonical_field_elements(good * 5 + bytes(32))
    k.helpers.validate_blob_data_as_canonical_field_elements(b"")
    with pytest.raises(k.errors.InvalidInputLength):
        k.helpers.validate_blob_data_as_canonical_field_elements(bytes(33))


# ---- through a context (GPU) ----------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_blob_to_polynomial(k):                                               # helpers_test.rs:390-428
    contents = open(os.path.join(GOLDEN, "blobs.txt"), "rb").read()
    read_fr_from_bytes = [int.from_bytes(contents[i:i + 32], "big") % R_ for i in range(0, len(contents), 32)]
    fr_from_str_vec = []
    for i, line in enumerate(open(os.path.join(GOLDEN, "blobs-from-fr.txt"))):
        fr_from_str = k.helpers.str_vec_to_fr_vec([line.rstrip().split(",")[0]])[0]
        fr_from_str_vec.append(fr_from_str)
        assert pyref.fr_from_mont(fr_from_str) == read_fr_from_bytes[i]
    got = k.helpers.blob_to_polynomial(contents)
    assert np.array_equal(np.stack(fr_from_str_vec), got)
    big = contents * (1 + (4096 * 32) // len(contents))                       # the same rows through `kzg_blob_to_fr` (the GPU codec takes over at 4 096 elements)
    got_big = k.helpers.blob_to_polynomial(big)
    want_big = pyref.frs_to_mont([int.from_bytes(big[i:i + 32].ljust(32, b"\0"), "big") % R_ for i in range(0, len(big), 32)])
    assert np.array_equal(got_big, want_big)


@pytest.mark.gpu
def test_compute_challenge_comprehensive(k):                                  # helpers_test.rs:847-949
    rng = random.Random(847)
    blob = k.Blob.from_raw_data(b"comprehensive test data for compute challenge validation")
    challenge = k.helpers.compute_challenge(blob, rand_g1(rng))
    assert pyref.fr_from_mont(challenge) != 0
    k.helpers.compute_challenge(blob, np.zeros(8, dtype=np.uint64))           # the identity is not rejected
    k.helpers.compute_challenge(blob, pyref.point_to_wire((1, 2)))            # nor the generator
    invalid = np.concatenate([pyref.fq_to_mont(1), pyref.fq_to_mont(1)])
    assert not k.helpers.is_on_curve_g1(invalid)
    with pytest.raises(k.errors.NotOnCurveError) as e:
        k.helpers.compute_challenge(blob, invalid)
    assert e.value.message == "G1 point not on curve"
    commitment = rand_g1(rng)
    c1, c2, c3 = (k.helpers.compute_challenge(blob, commitment) for _ in range(3))
    assert np.array_equal(c1, c2) and np.array_equal(c2, c3)
    assert np.array_equal(c1, k.helpers.compute_challenge_py(blob, commitment))                  # and equal to the transcript assembled in Python
    blob1, blob2 = k.Blob.from_raw_data(b"first test blob data"), k.Blob.from_raw_data(b"second test blob data")
    commitment1, commitment2 = rand_g1(rng), rand_g1(rng)
    c11, c21, c12 = k.helpers.compute_challenge(blob1, commitment1), k.helpers.compute_challenge(blob2, commitment1), k.helpers.compute_challenge(blob1, commitment2)
    assert not np.array_equal(c11, c21) and not np.array_equal(c11, c12)


@pytest.mark.gpu
def test_compute_challenges_and_evaluate_polynomial(k):                       # helpers_test.rs:952-1040
    rng = random.Random(952)
    blob1, blob2 = k.Blob.from_raw_data(b"test blob 1"), k.Blob.from_raw_data(b"test blob 2 with more data")
    commitment1, commitment2 = rand_g1(rng), rand_g1(rng)
    challenges, ys = k.helpers.compute_challenges_and_evaluate_polynomial([blob1, blob2], [commitment1, commitment2])
    assert len(challenges) == 2 and len(ys) == 2 and not np.array_equal(challenges[0], challenges[1])
    challenges, ys = k.helpers.compute_challenges_and_evaluate_polynomial([], [])
    assert len(challenges) == 0 and len(ys) == 0
    with pytest.raises(k.errors.GenericError) as e:
        k.helpers.compute_challenges_and_evaluate_polynomial([blob1, blob2], [commitment1])
    assert "length's of the input are not the same" in e.value.message
    with pytest.raises(k.errors.KzgError):
        k.helpers.compute_challenges_and_evaluate_polynomial([blob1], [commitment1, commitment2])
    challenges, ys = k.helpers.compute_challenges_and_evaluate_polynomial([blob2], [commitment2])
    assert len(challenges) == 1 and len(ys) == 1
    # the single result == the blob-by-blob composition (compute_challenge + evaluate_polynomial_in_evaluation_form)
    zs_py, ys_py = k.helpers.compute_challenges_and_evaluate_polynomial_py([blob2], [commitment2])
    assert np.array_equal(challenges[0], zs_py[0]) and np.array_equal(ys[0], ys_py[0])


@pytest.mark.gpu
def test_read_g1_point_from_bytes_be(k):                                      # helpers.rs:175-227 against the reference's g1.point <-> srs.g1.points.string
    raw = open(os.path.join(GOLDEN, "g1.point"), "rb").read()
    rows = [line.strip() for line in open(os.path.join(GOLDEN, "srs.g1.points.string")) if line.strip()]
    for i in (0, 1, 2, 17, 2999):
        got = k.helpers.read_g1_point_from_bytes_be(raw[32 * i:32 * i + 32])
        x, y = rows[i].strip("()").replace(" ", "").split(",")[:2]
        assert pyref.point_from_wire(got) == (int(x), int(y))
    assert not k.helpers.read_g1_point_from_bytes_be(bytes([0x40]) + bytes(31)).any()                  # the identity
    with pytest.raises(k.errors.DeserializationError, match="not enough bytes for g1 point"):
        k.helpers.read_g1_point_from_bytes_be(raw[:31])
    with pytest.raises(k.errors.DeserializationError, match="point at infinity not coded properly for g1"):
        k.helpers.read_g1_point_from_bytes_be(bytes([0x40]) + bytes(30) + b"\x01")
    bad = None
    for x in range(1, 50):                                                    # an x whose x^3 + 3 is no square
        if pow((x * x * x + 3) % P_, (P_ - 1) // 2, P_) != 1:
            bad = (0x80 << 248 | x).to_bytes(32, "big")
            break
    with pytest.raises(k.errors.NotOnCurveError, match="compressed g1 point not on curve"):
        k.helpers.read_g1_point_from_bytes_be(bad)


@pytest.mark.gpu
def test_srs_process_chunks_and_parallel_read(k):                             # srs.rs:51-70, :205-251
    path = os.path.join(GOLDEN, "g1.point")
    raw = open(path, "rb").read()
    pts = k.SRS.parallel_read_g1_points_native(path, 64, False)
    assert pts.shape == (64, 8)
    order = [5, 0, 63, 17]
    got = k.SRS.process_chunks((raw[32 * i:32 * i + 32], i, False) for i in order)
    assert [pos for _pt, pos in got] == order
    for (pt, pos) in got:
        assert np.array_equal(pt, pts[pos])
    # the arkworks-native compressed form of the same points (x little-endian, flags in the last byte) decodes to the same points
    native = [k.helpers.serialize_compressed(pts[i]) for i in order]
    got_n = k.SRS.process_chunks(zip(native, order, [True] * 4))
    for (pt, pos) in got_n:
        assert np.array_equal(pt, pts[pos])
    with pytest.raises(k.errors.DeserializationError, match="Failed to read point from bytes"):
        k.SRS.process_chunks([(b"\x40" + bytes(30) + b"\x01", 0, False)])


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [0, 1, 3, 10, 13])
def test_compute_quotient_eval_on_domain(k, log_n):                           # kzg.rs:237-260 against its literal restatement over big integers
    n = 1 << log_n
    rng = random.Random(237 + log_n)
    kzg = k.KZG.new()
    kzg.calculate_and_store_roots_of_unity(n * 32)
    w = pyref.root_of_unity(log_n)
    roots = [pow(w, i, R_) for i in range(n)]
    assert pyref.frs_from_mont(kzg.get_roots_of_unities()) == roots
    evals = [rng.randrange(R_) for _ in range(n)]

    def literal(z, value):
        q = 0
        for i, omega_i in enumerate(roots):
            if omega_i == z:
                continue
            q += (evals[i] - value) * omega_i * pow((z - omega_i) * z, R_ - 2, R_)
        return q % R_

    cases = [(roots[m], evals[m]) for m in sorted({0, n // 2, n - 1, rng.randrange(n)})]      # z = w^m, value = f_m: the use at kzg.rs:160-168
    cases.append((rng.randrange(1, R_), rng.randrange(R_)))                                    # z off the domain: nothing is skipped
    cases.append((roots[n // 3], rng.randrange(R_)))                                           # a domain point with a value that is not f_m
    for z, value in cases:
        got = kzg.compute_quotient_eval_on_domain(pyref.fr_to_mont(z), pyref.frs_to_mont(evals), pyref.fr_to_mont(value))
        assert pyref.fr_from_mont(got) == literal(z, value), (log_n, z)
    with pytest.raises(ZeroDivisionError):
        kzg.compute_quotient_eval_on_domain(pyref.fr_to_mont(0), pyref.frs_to_mont(evals), pyref.fr_to_mont(1))
    with pytest.raises(IndexError):
        kzg.compute_quotient_eval_on_domain(pyref.fr_to_mont(roots[0]), pyref.frs_to_mont(evals[:-1]) if n > 1 else np.zeros((0, 4), np.uint64), pyref.fr_to_mont(1))


@pytest.mark.gpu
def test_compute_quotient_eval_on_domain_2_18(k):                             # the same at 2^18 elements (many workgroups, the two-level sum): literal sum with one batched inversion
    log_n = 18
    n = 1 << log_n
    rng = random.Random(2018)
    kzg = k.KZG.new()
    kzg.calculate_and_store_roots_of_unity(n * 32)
    w = pyref.root_of_unity(log_n)
    roots = [1] * n
    for i in range(1, n):
        roots[i] = roots[i - 1] * w % R_
    evals = [rng.randrange(R_) for _ in range(n)]
    wire = pyref.frs_to_mont(evals)
    for m in (0, 12345, n - 1):
        z, value = roots[m], evals[m]
        dens = [(z - roots[i]) * z % R_ if i != m else 1 for i in range(n)]
        pref = [1] * (n + 1)
        for i in range(n):
            pref[i + 1] = pref[i] * dens[i] % R_
        inv_all = pow(pref[n], R_ - 2, R_)
        want = 0
        for i in range(n - 1, -1, -1):
            inv_i = inv_all * pref[i] % R_
            inv_all = inv_all * dens[i] % R_
            if i != m:
                want += (evals[i] - value) * roots[i] % R_ * inv_i
        got = kzg.compute_quotient_eval_on_domain(pyref.fr_to_mont(z), wire, pyref.fr_to_mont(value))
        assert pyref.fr_from_mont(got) == want % R_, m
