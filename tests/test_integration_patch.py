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
