"""Regenerates the transcript pin that integration/rust-kzg-bn254-mi355x.patch plants in the reference tree: constants produced by THIS
library's host-only entry points (kzg_compute_challenge, kzg_compute_r_powers; no GPU needed) for inputs any maintainer can rebuild from
ark-bn254 alone -- G = (1, 2), -G, the identity, a 62-byte raw blob -- written as two Rust test sources:
    primitives/tests/mi355x_transcript_pin.rs        helpers::compute_challenge + the three 32-byte `serialize_compressed` encodings
    verifier/src/batch.rs (a #[cfg(test)] module)    compute_r_powers of a 2-row batch (the function is private to that module)
The first `cargo test` with these files closes the one convention no reference vector pins: ark-serialize's flag bits.
Usage: python integration/make_transcript_pin.py  -> prints the two Rust sources between markers (tests/test_integration_patch.py compares
the constants in the patch with a fresh run of this module)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib, helpers

from rust_kzg_bn254_amd.consts import FQ_MODULUS as P
from rust_kzg_bn254_amd.fr import fq_from_int
RAW = b"kzg-bn254 on MI355X: transcript pin, sixty-two raw bytes long."
assert len(RAW) == 62
ZS, YS, LENS = (3, 5), (7, 11), (2, 4)


def point(x, y):
    return np.concatenate([fq_from_int(x), fq_from_int(y)]).astype(np.uint64)


def constants():
    lib = _lib.load()
    G, NEG = point(1, 2), point(1, P - 2)
    ident = np.zeros(8, np.uint64)
    data = helpers.pad_payload(RAW)                          # Blob::from_raw_data
    buf = np.frombuffer(data, dtype=np.uint8)
    out = {}
    for name, c in (("CHALLENGE_G", G), ("CHALLENGE_NEG_G", NEG), ("CHALLENGE_IDENTITY", ident)):
        z = np.zeros(4, np.uint64)
        rc = lib.kzg_compute_challenge(buf.ctypes.data_as(_lib.u8p), len(data), _lib.ptr(c), _lib.ptr(z))
        assert rc == 0, rc
        out[name] = str(k.fr.fr_to_int(z))
    for name, c in (("COMPRESSED_G", G), ("COMPRESSED_NEG_G", NEG), ("COMPRESSED_IDENTITY", ident)):
        out[name] = helpers.serialize_compressed(c)
    cm = np.ascontiguousarray(np.stack([G, NEG])); pf = np.ascontiguousarray(np.stack([NEG, G]))
    zs = np.ascontiguousarray(np.stack([k.fr.fr_from_int(v) for v in ZS])); ys = np.ascontiguousarray(np.stack([k.fr.fr_from_int(v) for v in YS]))
    lens = np.array(LENS, dtype=np.uint64)
    rp = np.zeros((2, 4), np.uint64)
    assert lib.kzg_compute_r_powers(_lib.ptr(cm), _lib.ptr(zs), _lib.ptr(ys), _lib.ptr(pf), _lib.ptr(lens), 2, _lib.ptr(rp)) == 0
    assert k.fr.fr_to_int(rp[0]) == 1
    out["R_POWER_1"] = str(k.fr.fr_to_int(rp[1]))
    return out


def rust_bytes(b):
    return "[" + ", ".join("0x%02x" % v for v in b) + "]"


def primitives_test(c):
    return '''//! Transcript pin planted by the MI355X patch (integration/make_transcript_pin.py of the kzg-bn254 MI355X library).
//!
//! The constants below were produced by `libkzg_bn254_mi355x.so` (kzg_compute_challenge and its ark-serialize restatement); this file uses
//! ONLY this workspace's own functions and arkworks.  If it passes, the library's Fiat-Shamir challenge -- in particular the flag bits of
//! `G1Affine::serialize_compressed`, which no test vector of this repository pins -- agrees with arkworks bit for bit.
use ark_bn254::{Fq, Fr, G1Affine};
use ark_ec::AffineRepr;
use ark_serialize::CanonicalSerialize;
use ark_std::str::FromStr;
use rust_kzg_bn254_primitives::{blob::Blob, helpers::compute_challenge};

const RAW: &[u8] = b"%(raw)s";
const CHALLENGE_G: &str = "%(CHALLENGE_G)s";
const CHALLENGE_NEG_G: &str = "%(CHALLENGE_NEG_G)s";
const CHALLENGE_IDENTITY: &str = "%(CHALLENGE_IDENTITY)s";
const COMPRESSED_G: [u8; 32] = %(COMPRESSED_G)s;
const COMPRESSED_NEG_G: [u8; 32] = %(COMPRESSED_NEG_G)s;
const COMPRESSED_IDENTITY: [u8; 32] = %(COMPRESSED_IDENTITY)s;

fn points() -> [G1Affine; 3] {
    let g = G1Affine::new_unchecked(Fq::from(1u64), Fq::from(2u64));
    let neg = G1Affine::new_unchecked(Fq::from(1u64), -Fq::from(2u64));
    [g, neg, G1Affine::identity()]
}

#[test]
fn mi355x_pin_serialize_compressed() {
    for (p, want) in points().iter().zip([COMPRESSED_G, COMPRESSED_NEG_G, COMPRESSED_IDENTITY]) {
        let mut v = Vec::new();
        p.serialize_compressed(&mut v).unwrap();
        assert_eq!(v, want.to_vec());
    }
}

#[test]
fn mi355x_pin_compute_challenge() {
    let blob = Blob::from_raw_data(RAW);
    for (p, want) in points().iter().zip([CHALLENGE_G, CHALLENGE_NEG_G, CHALLENGE_IDENTITY]) {
        assert_eq!(compute_challenge(&blob, p).unwrap(), Fr::from_str(want).unwrap());
    }
}
''' % dict(c, raw=RAW.decode(), COMPRESSED_G=rust_bytes(c["COMPRESSED_G"]), COMPRESSED_NEG_G=rust_bytes(c["COMPRESSED_NEG_G"]),
           COMPRESSED_IDENTITY=rust_bytes(c["COMPRESSED_IDENTITY"]))


def verifier_module(c):
    return '''
/// Transcript pin planted by the MI355X patch: `R_POWER_1` was produced by `libkzg_bn254_mi355x.so` (kzg_compute_r_powers) for the rows
/// below; this module uses only this file's own `compute_r_powers`.  See primitives/tests/mi355x_transcript_pin.rs.
#[cfg(test)]
mod mi355x_transcript_pin {
    use super::*;
    use ark_bn254::Fq;
    use ark_std::{str::FromStr, One};

    const R_POWER_1: &str = "%(R_POWER_1)s";

    #[test]
    fn mi355x_pin_compute_r_powers() {
        let g = G1Affine::new_unchecked(Fq::from(1u64), Fq::from(2u64));
        let neg = G1Affine::new_unchecked(Fq::from(1u64), -Fq::from(2u64));
        let r = compute_r_powers(&[g, neg], &[Fr::from(3u64), Fr::from(5u64)], &[Fr::from(7u64), Fr::from(11u64)], &[neg, g], &[2, 4]).unwrap();
        assert_eq!(r.len(), 2);
        assert!(r[0].is_one());
        assert_eq!(r[1], Fr::from_str(R_POWER_1).unwrap());
    }
}
''' % c


if __name__ == "__main__":
    c = constants()
    print("=== primitives/tests/mi355x_transcript_pin.rs ===")
    print(primitives_test(c), end="")
    print("=== verifier/src/batch.rs (appended) ===")
    print(verifier_module(c), end="")
