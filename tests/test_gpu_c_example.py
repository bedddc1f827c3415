"""-m gpu: the C-ABI used from plain C (examples/commit_and_verify.c), built with gcc against the in-tree library and run
as its own process: commit_blob, compute_proof, verify_proof against [tau]G2, the two-slot stream and the one-call batch verification
of 16 blobs — no Python, no torch
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
    assert "proof verifies" in res.stdout and "4 / 4" in res.stdout and "batch verifies" in res.stdout


def test_multi_device_c_example(tmp_path):
    """examples/multi_commit.c: kzg_multi_* from plain C, three contexts on GPU 0, folded commitment == single-device commitment."""
    exe = str(tmp_path / "multi_commit")
    libdir = os.path.join(ROOT, "rust-kzg-bn254_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "multi_commit.c"),
                           "-L" + libdir, "-lkzg_bn254_mi355x", "-Wl,-rpath," + libdir, "-o", exe])
    res = subprocess.run([exe, "0", "0", "0"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, (res.returncode, res.stdout, res.stderr)
    assert "3 device context(s)" in res.stdout and "== the single-device one" in res.stdout
