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
    assert "streamed commitments + blob proofs equal the one-call ones: 16 / 16" in res.stdout


def test_multi_device_c_example(tmp_path):
    """examples/multi_commit.c: kzg_multi_* from plain C, three contexts on GPU 0, folded commitment == single-device commitment."""
    exe = str(tmp_path / "multi_commit")
    libdir = os.path.join(ROOT, "rust-kzg-bn254_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "multi_commit.c"),
                           "-L" + libdir, "-lkzg_bn254_mi355x", "-Wl,-rpath," + libdir, "-o", exe])
    res = subprocess.run([exe, "0", "0", "0"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, (res.returncode, res.stdout, res.stderr)
    assert "3 device context(s)" in res.stdout and "== the single-device one" in res.stdout


def test_rccl_c_example_one_rank(tmp_path):
    """examples/rccl_commit.c: kzg_commit_coeff_form_rccl from plain C with a communicator the program creates itself (one rank on this
    one-GPU box): the RCCL path behind the C-ABI == the plain commitment."""
    exe = str(tmp_path / "rccl_commit")
    libdir = os.path.join(ROOT, "rust-kzg-bn254_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           os.path.join(ROOT, "examples", "rccl_commit.c"), "-L" + libdir, "-lkzg_bn254_mi355x", "-L/opt/rocm/lib", "-lrccl", "-lamdhip64",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(key, None)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, (res.returncode, res.stdout, res.stderr[-1500:])
    assert "one-rank RCCL commitment == the plain commitment" in res.stdout
    assert "one-rank config 4 over RCCL == the one-GPU commitment, proof and y" in res.stdout      # kzg_commit_eval_form_rccl / kzg_compute_proof_rccl
