"""The reference's own criterion benches (prover/benches/*.rs, verifier/benches/bench_kzg_verify.rs), same ids and input shapes,
run through this library's Python mirror on one MI355X — and, in the CPU column, through the oracle (C restatement of the
reference's algorithms, `threads` host threads; NOT arkworks).  SRS: 2^19 known-tau points generated on the device (the
reference loads g1.32mb.point, 524288 points).  Median of `reps` calls after 2 warm-ups; everything from host buffers, as
the reference's API hands them over.
Usage: python tools/bench_reference_suite.py [--cpu]      (--cpu adds the oracle column; the large MSMs take a few seconds)"""
import argparse, hashlib, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401  (load order: torch before the library)
import rust_kzg_bn254_amd as k

ap = argparse.ArgumentParser(); ap.add_argument("--cpu", action="store_true"); ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
FR = k.consts.FR_MODULUS
TAU = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % FR
rng = np.random.default_rng(20260101)


def med(fn, reps=args.reps):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return statistics.median(ts) * 1e3


def blob_of(nbytes):
    return k.Blob.from_raw_data(rng.integers(32, 127, size=nbytes, dtype=np.uint8).tobytes())


t0 = time.perf_counter()
srs = k.SRS.generate(TAU, 1 << 19)
print(f"SRS 2^19 generated + window tables in {(time.perf_counter() - t0) * 1e3:.0f} ms (the reference's SRS::new of 524288 points: 'a few minutes')")
kz = k.KZG.new()
g2_tau = k.helpers.g2_mul_generator(k.fr.fr_from_int(TAU))
orc = None
if args.cpu:
    import oracle as orc
    g1_host = srs.g1
rows = []

for nbytes in (10000, 30000, 50000):                                   # bench_kzg_commit.rs:17-42 (coefficient form)
    poly = blob_of(nbytes).to_polynomial_coeff_form()
    gpu = med(lambda: kz.commit_coeff_form(poly, srs))
    cpu = med(lambda: orc.commit_coeff_form(g1_host[:len(poly)], poly.coeffs()), 3) if orc else None
    rows.append((f"bench_kzg_commit_{nbytes}", f"n = {len(poly)}", gpu, cpu))
for nbytes, name in ((8_000_000, "8mb"), (16_252_000, "16mb")):     # bench_kzg_commit_large_blobs.rs:17-37
    poly = blob_of(nbytes).to_polynomial_coeff_form()
    gpu = med(lambda: kz.commit_coeff_form(poly, srs), 5)
    cpu = med(lambda: orc.commit_coeff_form(g1_host[:len(poly)], poly.coeffs()), 1) if orc else None
    rows.append((f"bench_kzg_commit_{name}", f"n = {len(poly)}", gpu, cpu))
for nbytes in (10000, 30000, 50000):                                   # bench_kzg_proof.rs:17-58
    blob = blob_of(nbytes); poly = blob.to_polynomial_eval_form()
    kz.calculate_and_store_roots_of_unity(len(blob))
    idx = int(rng.integers(0, poly.len_underlying_blob_field_elements()))
    gpu = med(lambda: kz.compute_proof_with_known_z_fr_index(poly, idx, srs))
    cpu = None
    if orc:
        rc, roots = orc.calculate_roots_of_unity(len(blob))
        cpu = med(lambda: orc.compute_proof(g1_host[:len(poly)], poly.evaluations(), roots, roots[idx], literal=False), 3)
    rows.append((f"bench_kzg_proof_{nbytes}", f"n = {len(poly)}, z = w^{idx}", gpu, cpu))
n = 1
while n <= 2048:                                                       # bench_g1_ifft.rs:17-32 (every power of two <= 2048)
    nn = n
    gpu = med(lambda: kz.g1_ifft(nn, srs), 5)
    cpu = med(lambda: orc.g1_ifft(g1_host[:nn], nn), 1) if (orc and nn <= 256) else None
    rows.append((f"bench_g1_ifft ({n})", "", gpu, cpu))
    n *= 2 if n < 64 else 4 if n < 1024 else 2
for nbytes in (10000, 30000, 50000):                                   # verifier/benches/bench_kzg_verify.rs:18-67
    blob = blob_of(nbytes); poly = blob.to_polynomial_eval_form()
    kz.calculate_and_store_roots_of_unity(len(blob))
    idx = int(rng.integers(0, poly.len_underlying_blob_field_elements()))
    commitment = kz.commit_eval_form(poly, srs)
    proof = kz.compute_proof_with_known_z_fr_index(poly, idx, srs)
    value, z = poly.get_evalualtion(idx), kz.get_nth_root_of_unity(idx)
    assert k.verify_proof(commitment, proof, value, z, g2_tau) is True
    rows.append((f"bench_kzg_verify_{nbytes}", "host pairing check (O(1))", med(lambda: k.verify_proof(commitment, proof, value, z, g2_tau)), None))

print("\n| criterion id (reference) | shape | this library, 1 x MI355X (ms) | oracle on the host CPU (ms) |")
print("|---|---|---|---|")
for name, shape, gpu, cpu in rows:
    print(f"| `{name}` | {shape} | {gpu:.3f} | {'' if cpu is None else f'{cpu:.1f}'} |")
