"""CPU: the reference-side change shipped as an artefact -- integration/rust-kzg-bn254-mi355x.patch (a `mi355x` cargo feature: new
primitives/src/mi355x.rs with the `extern "C"` block + the call-site replacements at prover/src/kzg.rs:100, :121, :141, :275-279,
primitives/src/polynomial.rs:131-135, :242-246, primitives/src/helpers.rs:332).  No Rust toolchain exists in the image, so what CAN be
checked is checked: the patch applies to the reference tree as it lies (`git apply --check`, in a scratch copy -- skipped where
/root/reference does not exist, e.g. on the GPU box), every `kzg_*` function its extern block declares is declared by the header with
the same number of parameters and exported by the built library, and every call site the header's top comment names is touched."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "integration", "rust-kzg-bn254-mi355x.patch")
REFERENCE = "/root/reference"


def added_lines():
    return [ln[1:] for ln in open(PATCH) if ln.startswith("+") and not ln.startswith("+++")]


def split_params(arglist):
    arglist = arglist.strip()
    if arglist in ("", "void"):
        return []
    return [a for a in arglist.split(",") if a.strip()]


def test_patch_applies_to_the_reference_tree(tmp_path):
    if not os.path.isdir(REFERENCE) or shutil.which("git") is None:
        pytest.skip("no /root/reference (or no git) on this machine")
    work = tmp_path / "ref"
    shutil.copytree(REFERENCE, work)
    subprocess.check_call(["git", "init", "-q", "."], cwd=work)
    res = subprocess.run(["git", "apply", "--check", "--verbose", PATCH], cwd=work, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    subprocess.check_call(["git", "apply", PATCH], cwd=work)
    rs = (work / "primitives" / "src" / "mi355x.rs").read_text()
    assert 'extern "C"' in rs and "#[link(name = \"kzg_bn254_mi355x\")]" in rs
    # feature-gated: without `--features mi355x` every original line is still compiled
    for rel in ("prover/src/kzg.rs", "primitives/src/polynomial.rs", "primitives/src/helpers.rs"):
        new, old = (work / rel).read_text(), open(os.path.join(REFERENCE, rel)).read()
        assert new.count('cfg(feature = "mi355x")') >= 1 and new.count('cfg(not(feature = "mi355x"))') >= 1, rel
        kept = [ln for ln in old.splitlines() if ln.strip()]
        assert all(ln in new for ln in kept), rel                  # nothing of the reference was deleted, only gated
    assert 'mi355x = ["rust-kzg-bn254-primitives/mi355x"]' in (work / "prover" / "Cargo.toml").read_text()


def test_extern_block_matches_the_header_and_the_library():
    import sys
    sys.path.insert(0, ROOT)
    import rust_kzg_bn254_amd as k
    lib = k._lib.load()
    text = "".join(added_lines())
    block = text[text.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    rust = {m.group(1): len(split_params(m.group(2))) for m in re.finditer(r"fn (kzg_[a-z0-9_]+)\(([^)]*)\)", block)}
    assert len(rust) >= 12, rust
    hdr = open(os.path.join(ROOT, "include", "kzg_bn254_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    c = {m.group(1): len(split_params(m.group(2))) for m in re.finditer(r"\b(kzg_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", hdr)}
    for name, n_args in rust.items():
        assert name in c, "%s is declared by the patch but not by the header" % name
        assert c[name] == n_args, "%s: %d parameters in the patch, %d in the header" % (name, n_args, c[name])
        assert hasattr(lib, name), "%s is not exported by the library" % name
        assert len(k._lib.PROTOTYPES[name][1]) == n_args, name      # and the ctypes mirror agrees


def test_patch_touches_the_call_sites_of_the_boundary():
    """SURVEY §8b / DESIGN §1: the five arkworks call sites (+ compute_proof_impl as one call)."""
    text = open(PATCH).read()
    files = set(re.findall(r"^\+\+\+ b/(\S+)", text, flags=re.M))
    assert {"prover/src/kzg.rs", "primitives/src/polynomial.rs", "primitives/src/helpers.rs", "primitives/src/mi355x.rs", "primitives/src/lib.rs",
            "primitives/Cargo.toml", "prover/Cargo.toml"} <= files
    plus = "".join(added_lines())
    for call in ("mi355x::commit_eval_form(", "mi355x::commit_coeff_form(", "mi355x::g1_ifft(", "mi355x::compute_proof(", "mi355x::fr_ntt(&self.evaluations, true)",
                 "mi355x::fr_ntt(&self.coeffs, false)", "mi355x::msm(points, scalars)"):
        assert call in plus, call


def test_transcript_pin_in_the_patch_is_what_the_library_computes():
    """the one convention no reference vector pins -- ark-serialize's compressed G1 flags inside compute_challenge /
    compute_r_powers -- is planted in the reference tree as Rust tests whose constants come from THIS library (host-only entry points, no
    GPU).  Here: the constants in the patch equal a fresh run (integration/make_transcript_pin.py), and the test sources use nothing but the
    reference's own functions, so the first `cargo test` of a maintainer closes the pin."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "integration"))
    import make_transcript_pin as pin
    c = pin.constants()
    plus = "".join(added_lines())
    for name in ("CHALLENGE_G", "CHALLENGE_NEG_G", "CHALLENGE_IDENTITY", "R_POWER_1"):
        m = re.search(r'const %s: &str = "(\d+)";' % name, plus)
        assert m and m.group(1) == c[name], name
    for name in ("COMPRESSED_G", "COMPRESSED_NEG_G", "COMPRESSED_IDENTITY"):
        m = re.search(r"const %s: \[u8; 32\] = \[([^\]]*)\];" % name, plus)
        assert m and bytes(int(v, 16) for v in m.group(1).split(",")) == c[name], name
    # the three encodings cover both flag bits and the no-flag case; the challenges differ pairwise, so a wrong flag cannot go unnoticed
    assert c["COMPRESSED_G"][31] == 0x00 and c["COMPRESSED_NEG_G"][31] == 0x80 and c["COMPRESSED_IDENTITY"][31] == 0x40
    assert len({c["CHALLENGE_G"], c["CHALLENGE_NEG_G"], c["CHALLENGE_IDENTITY"]}) == 3
    assert pin.primitives_test(c) in plus.replace("\r", "") or all(ln in plus for ln in pin.primitives_test(c).splitlines() if ln.strip())
    assert "fn mi355x_pin_compute_r_powers()" in plus and "primitives/tests/mi355x_transcript_pin.rs" in open(PATCH).read()
    # independent of the library: the same transcript through hashlib and the Python mirror's serialisation (layout slips)
    import hashlib
    import rust_kzg_bn254_amd as k
    from rust_kzg_bn254_amd import helpers
    data = helpers.pad_payload(pin.RAW)
    n = 1
    while n < len(data) // 32:
        n <<= 1
    msg = b"EIGENDA_FSBLOBVERIFY_V1_" + n.to_bytes(8, "big") + data + bytes(32 * n - len(data)) + c["COMPRESSED_NEG_G"]
    assert str(int.from_bytes(hashlib.sha256(msg).digest(), "big") % k.consts.FR_MODULUS) == c["CHALLENGE_NEG_G"]
