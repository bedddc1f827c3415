"""30 proofs (z on the domain) and 30 commitments of a 2048-evaluation polynomial on a 2^19-point SRS, as the reference's bench_kzg_proof /
bench_kzg_commit do: for rocprofv3 --kernel-trace --stats (which kernels make up a small proof)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hashlib
import numpy as np, torch, bench
import rust_kzg_bn254_amd as k
tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % bench.FR
srs = k.SRS.generate(tau, 1 << int(os.environ.get("SRS_LOG", "19")))
kz = k.KZG.new()
nbytes = int(os.environ.get("NBYTES", "50000"))
blob = k.Blob.from_raw_data(bytes((i * 7 + 3) % 251 for i in range(nbytes)))
poly = blob.to_polynomial_eval_form()
kz.calculate_and_store_roots_of_unity(len(blob))
def med(f, reps=30):
    for _ in range(3): f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return sorted(ts)[len(ts) // 2] * 1e3
print("n = %d" % len(poly))
print("proof  %.3f ms" % med(lambda: kz.compute_proof_with_known_z_fr_index(poly, 226, srs)), flush=True)
print("commit %.3f ms" % med(lambda: kz.commit_eval_form(poly, srs)), flush=True)
