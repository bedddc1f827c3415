"""-m gpu: KZG_BATCH_C (bucket bits per polynomial of the batched commitments, read once per process) forced to 13 / 14 / 15 / 16 and to the refused 12
(c = 12 would be zero units of 4 096 buckets per polynomial -- silently wrong commitments): every batched result must equal the single-call
result, at a short (2^9) and a long (2^14) polynomial length.  One child process per value."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, hashlib, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import torch  # noqa: F401
import pyref
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.default_context(); P = _lib.ptr
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") %% pyref.R_
srs = k.SRS.generate(tau, 1 << 14)
rng = np.random.default_rng(12)
for n, count in ((1 << 9, 24), (1 << 14, 5)):
    sc = np.zeros((count * n, 4), dtype=np.uint64)
    sc[:, :3] = rng.integers(0, 2**63, size=(count * n, 3), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(count * n, 3), dtype=np.uint64)
    sc[:, 3] = rng.integers(0, 2**60, size=count * n, dtype=np.uint64)         # < r: canonical Montgomery words
    out = np.zeros((count, 8), dtype=np.uint64)
    assert lib.kzg_commit_coeff_form_batch(ctx.handle, srs.handle, P(sc), n, count, P(out), None) == 0
    one = np.zeros(8, dtype=np.uint64); inf = C.c_uint8(0)
    for j in range(count):
        row = np.ascontiguousarray(sc[j * n:(j + 1) * n])
        assert lib.kzg_commit_coeff_form(ctx.handle, srs.handle, P(row), n, P(one), C.byref(inf)) == 0
        assert np.array_equal(one, out[j]), (n, j)
print("batched == single")
"""


@pytest.mark.parametrize("forced", ["12", "13", "14", "15", "16"])
def test_forced_batch_bucket_bits_match_the_single_calls(forced):
    env = dict(os.environ, KZG_BATCH_C=forced)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "batched == single" in r.stdout, r.stdout[-500:] + r.stderr[-1500:]
