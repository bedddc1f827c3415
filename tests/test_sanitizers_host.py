"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer builds (g++ / gcc) of everything in this repo that runs on the host without a
GPU: the product's host headers (host_pairing.h, host_curve.h, host_sha256.h), the host form of the device math headers
(field29.h / curve.h with the lazy-reduction bound checks on) and the oracle.  The reference's CI runs its whole suite on two targets
(/root/reference/.github/workflows/rust.yml:35-49); GPU sanitizers are not available on this pool, so the device code is covered by
its host form here and by the parity tests on the GPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "rust-kzg-bn254_amd", "csrc")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def test_pairing_self_check_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "pairingcheck_san")
    subprocess.check_call(["g++", "-std=c++17", *SAN, "-I" + CSRC, os.path.join(HERE, "hostcheck", "pairingcheck.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    props = [ln.split() for ln in r.stdout.splitlines() if len(ln.split()) == 2 and ln.split()[1] in ("0", "1")]
    assert len(props) >= 25 and all(v == "1" for _, v in props), [p for p in props if p[1] != "1"]


def test_device_math_host_form_sha256_fold_and_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "sanitize_main")
    objs = []
    for src in ("kzg_oracle.c", "field.c"):
        o = str(tmp_path / (src + ".o"))
        subprocess.check_call(["gcc", "-std=gnu11", *SAN, "-c", os.path.join(ROOT, "oracle", src), "-o", o])
        objs.append(o)
    subprocess.check_call(["g++", "-std=c++17", *SAN, "-DKZG_BOUND_CHECK", "-Wno-unknown-pragmas", "-I" + CSRC, "-I" + os.path.join(HERE, "hostcheck"),
                           os.path.join(HERE, "hostcheck", "sanitize_main.cpp"), *objs, "-lpthread", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0 and "sanitize ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_transcript_generator_and_two_stream_sha_under_asan_ubsan(tmp_path):
    """csrc/host_transcript.h + host_sha256.h (round 6: the transcript prefix as segments, two SHA-256 streams interleaved): 300 pairs of exactly-sized buffers."""
    exe = str(tmp_path / "transcriptcheck_san")
    subprocess.check_call(["g++", "-std=c++17", *SAN, "-DTRANSCRIPT_SELF_CHECK", "-I" + CSRC, os.path.join(HERE, "hostcheck", "transcriptcheck.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0 and "transcript self-check ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
