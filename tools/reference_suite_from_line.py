"""profiles/rNN_reference_bench_suite.md FROM a bench.py line (one source of numbers): every criterion harness of the
reference (prover/benches/bench_kzg_{setup,commit,commit_large_blobs,proof}.rs, bench_g1_ifft.rs, verifier/benches/bench_kzg_verify.rs)
with the figure the driver-run line carries for its shape and, where the line has it, the oracle port on the box's host cores beside it.
Usage: python tools/reference_suite_from_line.py profiles/r06_bench_line.json > profiles/r06_reference_bench_suite.md"""
import json
import sys

d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][0])
sec = d["secondary"]
sh, st = sec["reference_bench_shapes"], sec["reference_bench_shapes_stats"]
cpu = sec.get("reference_bench_shapes_cpu_port", {})


def row(cid, shape, key, cpu_key=None, note=""):
    v = sh.get(key + "_ms")
    s = st.get(key, {})
    c = cpu.get(cpu_key) if cpu_key else None
    print("| `%s` | %s | %s | %s | %s | %s |" % (cid, shape, "%.3f" % v if v is not None else "—", "%.3f" % s["p99"] if "p99" in s else ("%.3f" % s["max"] if "max" in s else "—"),
                                              ("%.0f" % c if c and c >= 100 else "%.1f" % c) if c else "—", note))


print("# The reference's criterion suite on one MI355X, from the bench line `%s`" % sys.argv[1])
print()
print("Host buffers in, host results out, one call at a time (the harnesses' shape); median of the per-call wall times (`reference_bench_shapes`), p99 / max beside it;")
print("CPU column: the oracle port (`oracle/`, NOT arkworks) on the box's host cores in the same run (`reference_bench_shapes_cpu_port`; %s)." % cpu.get("commit_is", "—"))
print()
print("| criterion id (reference) | shape | this library (ms, median) | p99 / max (ms) | oracle port on the host (ms) | note |")
print("|---|---|---|---|---|---|")
setup_key = [k for k in sh if k.startswith("kzg_setup_") and k.endswith("_ms")]
if setup_key:
    k0 = setup_key[0][:-3]
    row("bench_kzg_setup", "`SRS::new(.., 524288)`: 16 MiB of compressed points", k0, k0 + "_ms", cpu.get("kzg_setup_is", ""))
for nbytes, nn in ((10000, 512), (30000, 1024), (50000, 2048)):
    row("bench_kzg_commit_%d" % nbytes, "n = %d coefficients" % nn, "commit_coeff_%d" % nn)
row("bench_kzg_commit_8mb", "258 065 coefficients in 2^18", "commit_8mb", "commit_8mb_ms")
row("bench_kzg_commit_16mb", "524 259 coefficients in 2^19", "commit_16mb", "commit_16mb_ms")
for nbytes, nn in ((10000, 512), (30000, 1024), (50000, 2048)):
    row("bench_kzg_proof_%d" % nbytes, "n = %d, z a domain point" % nn, "compute_proof_%d" % nn, note="off the domain: %.3f ms" % sh["compute_proof_off_domain_%d_ms" % nn])
for nn in (512, 1024, 2048):
    row("bench_g1_ifft (%d)" % nn, "", "g1_ifft_%d" % nn)
row("bench_kzg_verify_*", "one `verify_proof` (host pairing check, O(1))", "verify_proof")
print()
print("Beyond the harnesses, same line: blob -> commitment + proof streamed %.2f ms per 32 MiB blob (12 jobs in flight; one call at a time %.1f ms); batch verification of 4 096 blobs "
      "median %.2f ms, max %.2f over %d calls; 2^20-pair MSM step %.4f ms." % (sec["commit_and_prove_blob_streamed_ms"], sec["commit_and_prove_blob_from_host_bytes_ms"],
                                                                              sec["batch_verify_4096_end_to_end_stats"]["median"], sec["batch_verify_4096_end_to_end_stats"]["max"],
                                                                              sec["batch_verify_4096_end_to_end_stats"]["calls"], d["ms_per_step"]))
