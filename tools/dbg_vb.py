import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, torch
import rust_kzg_bn254_amd as k, oracle as orc, pyref
k.load(); k.default_context()
rng = np.random.default_rng(77)
def rb(n_raw): return k.Blob.from_raw_data(rng.integers(32, 127, size=n_raw, dtype=np.uint8).tobytes())
blobs = [rb(n) for n in (1, 31, 32, 62, 100, 300, 700, 1500, 3000, 7000, 15000, 31000, 50000, 63000, 126000, 127000)]
cms = [np.array(pyref.point_to_wire(pyref.ec_mul(1000 + i, (1, 2))), dtype=np.uint64) for i in range(len(blobs))]
zs, ys = k.helpers.compute_challenges_and_evaluate_polynomial(blobs, cms)
rc, zw, yw = orc.compute_challenges_and_evaluate_polynomial([b.data() for b in blobs], np.stack(cms))
for i, b in enumerate(blobs):
    print(i, len(b.to_polynomial_eval_form()), np.array_equal(zs[i], zw[i]), np.array_equal(ys[i], yw[i]))
