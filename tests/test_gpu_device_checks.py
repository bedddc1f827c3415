"""-m gpu: stand-alone device checks of the lane-quad point arithmetic (csrc/curve_quad.h) against the one-lane formulas of csrc/curve.h:
`quad_add`, `quad_madd` and (round 4) `quad_dbl` on 4 096 cases each, with the identity, doubling and cancellation cases mixed into every
wave (tools/ubench/quad_check.hip, built here with hipcc for gfx950).  The kernels that use them (reduction levels, bit sums, the g1_ifft
quad stage) are covered end to end elsewhere; this pins the building blocks themselves."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lane_quad_point_arithmetic_matches_the_one_lane_formulas(tmp_path):
    exe = str(tmp_path / "quad_check")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", os.path.join(ROOT, "tools", "ubench", "quad_check.hip"), "-o", exe],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-1500:]
    lines = res.stdout.strip().splitlines()
    assert lines[-1] == "OK" and len(lines) == 4
    for name, ln in zip(("quad_add", "quad_madd", "quad_dbl"), lines[:3]):
        assert ln.startswith(name + ": 0 of 4096 cases differ"), ln
