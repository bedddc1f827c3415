"""-m gpu: the C-ABI used from plain C (examples/commit_and_verify.c), built with gcc against the in-tree library and run
as its own process: commit_blob, compute_proof, verify_proof against [tau]G2, and the two-slot stream — no Python, no torch
on the product side."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_example_builds_and_runs(tmp_path):
    exe = str(tmp_path / "commit_and_verify")
    libdir = os.path.join(ROOT, "rust-kzg-bn254_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "commit_and_verify.c"),
                           "-L" + libdir, "-lkzg_bn254_mi355x", "-Wl,-rpath," + libdir, "-o", exe])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, (res.returncode, res.stdout, res.stderr)
    assert "proof verifies" in res.stdout and "4 / 4" in res.stdout
