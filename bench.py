#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X KZG-BN254 hot path.

Metric (BASELINE.json): KZG commitments/s + G1-MSM (scalar, point) pairs/s on a 2^20-point SRS.
A STEP is one coefficient-form commitment (`KZG::commit_coeff_form`, prover/src/kzg.rs:107-125 — what the
reference's `bench_kzg_commit_large_blobs` times): one G1 MSM of 2^20 scalars over the device-resident SRS,
scalars already in HBM when the timed region starts, result = affine G1 point on the host.

N = 1 : configs[1] "G1 MSM 2^20 scalars on 1 MI355X".
N > 1 : the same 2^20-pair MSM sharded by scalar index over N ranks (one process per GPU), one RCCL all-gather
        of the N partial sums (16 x u64 each) + host fold on every rank  ->  "scaling": "strong".

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel k_msm_accumulate, HIP events recorded by the
library on its own launch stream) and `cpu_baseline` (the oracle's arkworks-shaped Pippenger on the host cores).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

FR = 21888242871839275222246405745257275088548364400416034343698204186575808495617
LOG_N = int(os.environ.get("KZG_BENCH_LOG_N", "20"))
DEPTH = int(os.environ["KZG_BENCH_DEPTH"]) if os.environ.get("KZG_BENCH_DEPTH") else None    # MSMs in flight (default: sharding.py)
HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy ceiling is ~6.3 TB/s
BYTES_PER_PAIR = 96            # 64 B packed affine point + 32 B scalar, each read once
PMC_JSON = os.path.join(ROOT, "profiles", "r06_pmc_summary.json")
MADS_PER_MIXED_ADD = 1467      # v_mad_i64_i32 per xyzz_madd (8 x 162 + 2 x 126 - 81 for the fused Y3)
# The instruction stream of one mixed addition on the fast path of k_msm_accumulate's loop, by class: v_mad_i64_i32, v_mul_lo_u32, 64-bit
# shifts, v_and_b32, every other instruction (32-bit VALU / SALU / loads), s_nop.  REGENERATED from the compiler's listing by
# tools/count_isa.py (`hipcc -S` of csrc/msm.hip); tests/test_isa_counts.py fails when these constants and the listing disagree.  Priced
# with the issue rates of THIS box, measured in set-up at the kernel's occupancy (kzg_ctx_measure_valu_rates, three waves per SIMD).
VALU_MIX_COUNTS = (1467, 81, 154, 179, 227, 5)
VALU_RATES_FALLBACK_NS = (2.13, 2.03, 1.85, 1.24, 1.24, 0.42)     # profiles/r01_valu_rates_mi355x.txt, four waves per SIMD
N_SIMDS = 1024
N_BUFFERS = 8                  # distinct resident scalar buffers the timed steps rotate through


def cgroup_nr_throttled():
    """nr_throttled of this process's cgroup (v2 /sys/fs/cgroup/cpu.stat, v1 under cpu/), None where there is none: how many CFS periods froze the process so far."""
    for base in ("/sys/fs/cgroup", "/sys/fs/cgroup/cpu"):
        try:
            for ln in open(os.path.join(base, "cpu.stat")):
                f = ln.split()
                if len(f) == 2 and f[0] == "nr_throttled":
                    return int(f[1])
        except OSError:
            pass
    return None


def runqueue_wait_ns_by_thread():
    """/proc/self/task/<tid>/schedstat, second field: ns each thread of this process has spent RUNNABLE BUT WAITING for a CPU.  {} where the kernel does not keep it."""
    out = {}
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                out[tid] = int(open("/proc/self/task/%s/schedstat" % tid).read().split()[1])
            except (OSError, ValueError, IndexError):
                pass
    except OSError:
        pass
    return out


def host_pool_threads():
    """Threads of the library's host pool (capi.hip host_threads_cap): 48, the hardware threads, or the cgroup's CPU quota, whichever is least."""
    if os.environ.get("KZG_HOST_THREADS_MAX"):
        return int(os.environ["KZG_HOST_THREADS_MAX"])
    cap = min(48, os.cpu_count() or 1)
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            cap = min(cap, max(2, int(float(a) / float(b))))
    except (OSError, ValueError):
        pass
    return cap


def pmc_traffic_bytes(log_n):
    """(HBM bytes per k_msm_accumulate launch, reason): from the committed rocprofv3 --pmc passes of this round (FETCH_SIZE +
    WRITE_SIZE, KB; tools/pmc_summarize.py documents the gfx950 correction).  Only valid for the workload it was measured on."""
    try:
        d = json.load(open(PMC_JSON))
    except Exception as e:
        return None, "no PMC summary committed for this round (%s)" % type(e).__name__
    if d.get("log_n") != log_n:
        return None, "the committed PMC passes were collected at 2^%s pairs, this run is 2^%d" % (d.get("log_n"), log_n)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from pmc_summarize import kernel_sources_sha256
        now = kernel_sources_sha256(ROOT)
    except Exception as e:                                          # noqa: BLE001
        return None, "cannot digest the kernel sources (%s)" % type(e).__name__
    if d.get("kernel_sources_sha256") != now:
        return None, ("STALE: %s was collected on other kernel sources (sha256 %s..., the tree has %s...): re-run tools/collect_profiles.sh"
                      % (os.path.basename(PMC_JSON), str(d.get("kernel_sources_sha256"))[:12], now[:12]))
    a = d["kernels"]["k_msm_accumulate"]
    return (a["FETCH_SIZE_KB"] + a["WRITE_SIZE_KB"]) * 1024.0, "profiles/" + os.path.basename(PMC_JSON)


def ints_to_wire(vals):
    """canonical python ints -> (n, 4) uint64 wire array (Montgomery, R = 2^256)."""
    R = (1 << 256) % FR
    buf = b"".join((v * R % FR).to_bytes(32, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def uniform_scalars(n, seed):
    """Scalars-B of SURVEY.md 8(d): i.i.d. uniform in [0, r) (models quotient polynomials and verifier r-powers).
    Returns (canonical ints, wire array)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    vals = []
    while len(vals) < n:
        raw = rng.integers(0, 1 << 63, size=(n - len(vals) + 1024, 5), dtype=np.uint64)
        for a, b, c, d, e in raw.tolist():
            v = (a | (b << 63) | (c << 126) | (d << 189) | (e << 252)) & ((1 << 254) - 1)       # rejection sampling of 254 bits
            if v < FR:
                vals.append(v)
    vals = vals[:n]
    return vals, ints_to_wire(vals)


def expected_commitment(canonical_scalars, tau):
    """Known-tau identity: sum_i c_i [tau^i] G1 == (sum_i c_i tau^i mod r) * G1 -- big-integer arithmetic plus ONE
    scalar multiplication by an independent code path (tests/pyref.py, affine double-and-add); wire point (8 x u64)."""
    import pyref
    acc, cur = 0, 1
    for v in canonical_scalars:
        acc = (acc + v * cur) % FR
        cur = cur * tau % FR
    return pyref.point_to_wire(pyref.ec_mul(acc, (1, 2)))


def eval_form_ground_truth(evals, log_n, points):
    """f^(x) for every x in `points` (none on the domain) of the polynomial with the canonical evaluations `evals` on the 2^log_n-point
    domain w^i, w = 5^((r-1)/n): barycentric formula (primitives/src/helpers.rs:507-532) in python big integers -- the independent check
    of BASELINE config 4's commitment (x = tau) and proof (x = z)."""
    n = 1 << log_n
    w = pow(5, (FR - 1) >> log_n, FR)
    roots, cur = [], 1
    for _ in range(n):
        roots.append(cur); cur = cur * w % FR
    out = []
    for x in points:
        pre, acc = [], 1
        for r_ in roots:
            pre.append(acc); acc = acc * ((x - r_) % FR) % FR
        inv = pow(acc, -1, FR)
        s_ = 0
        for i in range(n - 1, -1, -1):
            iv = inv * pre[i] % FR
            inv = inv * ((x - roots[i]) % FR) % FR
            s_ += evals[i] * roots[i] % FR * iv
        out.append(s_ % FR * (pow(x, n, FR) - 1) % FR * pow(n, -1, FR) % FR)
    return out


def blob_like_canonical(n, seed):
    """Scalars-A of SURVEY.md 8(d): raw bytes uniform in [32,126] (bench_kzg_commit.rs:18), 31 per element behind
    a zero byte (helpers.rs:823-840)  ->  canonical values < 2^248 (python ints)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    raw = rng.integers(32, 127, size=(n, 31), dtype=np.uint8).tobytes()
    return [int.from_bytes(raw[31 * i:31 * i + 31], "big") for i in range(n)]


def blob_like_scalars(n, seed):
    """Scalars-A in wire (Montgomery) form."""
    return ints_to_wire(blob_like_canonical(n, seed))


def main_multi(args):
    """`bench.py --multi --gpus N`: one process, N devices behind one kzg_multi handle (csrc/multi.hip): device g holds SRS powers and
    scalars [g n / N, (g+1) n / N) resident, runs its own pipeline of partial MSMs on one host thread of the library, the host folds the
    N partial sums of every step.  Same workload, same rotation of resident buffers and the same per-step check as the default mode."""
    import rust_kzg_bn254_amd as k
    from rust_kzg_bn254_amd.sharding import MultiKzg
    n = 1 << LOG_N
    tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % FR
    have = torch.cuda.device_count()
    ids = list(range(args.gpus)) if have >= args.gpus else [i % max(have, 1) for i in range(args.gpus)]   # fewer devices than asked: several contexts per GPU (rehearsal)
    mk = MultiKzg(ids)
    mk.srs_generate(tau, n)
    seed0 = 0x4B5A472D424E3235 & 0x7FFFFFFF
    canon_sets = [blob_like_canonical(n, seed0 if j == 0 else seed0 + 1 + j) for j in range(N_BUFFERS // 2)]
    canon_sets += [c[-(n // 2 + j):] + c[:-(n // 2 + j)] for j, c in enumerate(canon_sets[:N_BUFFERS // 2])]
    for j, c in enumerate(canon_sets):
        mk.scalars_upload(j, ints_to_wire(c))
    rot = lambda count, start=0: [(start + i) % N_BUFFERS for i in range(count)]           # noqa: E731
    mk.commit_resident_stream(rot(4))                                                        # workspaces
    mk.commit_resident_stream(rot(min(48 * args.gpus, 384)))                                 # clock ramp (as the default mode)
    mk.commit_resident_stream(rot(args.warmup))
    for d in set(ids):
        torch.cuda.synchronize(d)
    t0 = time.perf_counter()
    res = mk.commit_resident_stream(rot(args.steps))
    for d in set(ids):
        torch.cuda.synchronize(d)
    elapsed = time.perf_counter() - t0
    wants = [expected_commitment(c, tau) for c in canon_sets]
    exact = bool(all(np.array_equal(res[i], wants[i % N_BUFFERS]) for i in range(args.steps)))
    out = {
        "metric": "G1-MSM (scalar,point) pairs/s = 2^%d x KZG coeff-form commitments/s, 2^%d-point SRS" % (LOG_N, LOG_N),
        "value": n * args.steps / elapsed, "unit": "pairs/s", "commitments_per_s": args.steps / elapsed,
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32x9 (29-bit limbs, 64-bit accumulate)",
        "data": "synthetic: SRS P_i = tau^i G1 with known tau (generated on the devices); blob-like scalars < 2^248 (Scalars-A); seeded",
        "config": {"workload": "G1 MSM 2^%d scalars (KZG::commit_coeff_form), scalars resident in HBM; step k commits buffer k mod %d" % (LOG_N, N_BUFFERS),
                   "sharding": "--multi: ONE process, %d device(s) %s behind one kzg_multi handle (one host thread and one software pipeline per "
                               "device, no collective; host fold of the partial sums)" % (args.gpus, ids),
                   "bit_exact_vs_oracle": exact,
                   "bit_exact_check": "every timed step's commitment == (sum_i c_i tau^i mod r) * G1 of its buffer (big integers + tests/pyref.py)"},
        "roofline": {"bound": "hbm", "achieved": BYTES_PER_PAIR * n * args.steps / elapsed / 1e9, "peak": HBM_PEAK_GBS * args.gpus, "unit": "GB/s",
                     "frac": BYTES_PER_PAIR * n * args.steps / elapsed / 1e9 / (HBM_PEAK_GBS * args.gpus), "traffic": None,
                     "kernel": "whole step (the per-kernel figures are in the default mode's line)"},
    }
    print(json.dumps(out), flush=True)
    mk.close()
    if not exact:
        raise SystemExit("bench.py --multi: a commitment differs from the expected point")


class ClockSampler:
    """Engine clock (and socket power) of THIS rank's GPU under the bench's load, read from sysfs every ~8 ms by a side thread:
    /sys/class/drm/cardN/device/pp_dpm_sclk marks the current level with '*'.  Box-to-box differences of the headline (1.08 .. 1.14 ms per step
    on the same code) should show here; nothing is set, only read.  Every failure (no sysfs, no match) leaves the report None."""

    def __init__(self, device_index):
        self.sclk_path = self.power_path = None
        self.samples, self.power = [], []
        self._stop = None
        try:
            import glob
            props = torch.cuda.get_device_properties(device_index)
            want = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, props.pci_device_id)
            for dev in glob.glob("/sys/class/drm/card*/device"):
                if os.path.basename(os.path.realpath(dev)) == want and os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
                    self.sclk_path = os.path.join(dev, "pp_dpm_sclk")
                    hw = glob.glob(os.path.join(dev, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(dev, "hwmon", "hwmon*", "power1_input"))
                    self.power_path = hw[0] if hw else None
        except Exception:                                           # noqa: BLE001
            self.sclk_path = None

    def _read(self):
        try:
            for ln in open(self.sclk_path):
                if "*" in ln:
                    self.samples.append(float(ln.split(":")[1].lower().split("mhz")[0]))
            if self.power_path:
                self.power.append(float(open(self.power_path).read()) / 1e6)
        except Exception:                                           # noqa: BLE001
            pass

    def __enter__(self):
        if self.sclk_path:
            import threading
            self._stop = threading.Event()

            def loop():
                while not self._stop.wait(0.008):
                    self._read()
            self._thread = threading.Thread(target=loop, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        if self._stop is not None:
            self._stop.set()
            self._thread.join()

    def report(self):
        if not self.samples:
            return None
        r = {"sclk_mhz_mean": sum(self.samples) / len(self.samples), "sclk_mhz_min": min(self.samples), "sclk_mhz_max": max(self.samples),
             "samples": len(self.samples), "source": "sysfs pp_dpm_sclk of this GPU, every ~8 ms during the untimed spin-up steps (the same pipelined load as the timed region)"}
        if self.power:
            r["socket_power_w_mean"] = sum(self.power) / len(self.power)
        return r


def visible_gpu_count():
    """GPUs this process may use, WITHOUT touching HIP (the launcher must stay free of a GPU runtime): the render nodes it can open
    (/dev/dri/renderD*: a container that is given one GPU has exactly one) and the KFD topology nodes with SIMDs it can read (the other
    GPUs of the host are not readable from inside such a container), narrowed by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES when set.  None when neither source exists (the ranks then fail fast on a missing device themselves)."""
    import glob
    counts = []
    nodes = glob.glob("/dev/dri/renderD*")
    if nodes or os.path.isdir("/dev/dri"):
        counts.append(sum(1 for p_ in nodes if os.access(p_, os.R_OK | os.W_OK)))
    kfd_nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if kfd_nodes:
        gpus = 0
        for props in kfd_nodes:
            try:
                for ln in open(props):
                    if ln.startswith("simd_count") and int(ln.split()[1]) > 0:
                        gpus += 1
            except OSError:                                     # a GPU of the host this container was not given
                pass
        counts.append(gpus)
    if not counts:
        return None
    gpus = min(counts)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            gpus = min(gpus, len([x for x in v.split(",") if x.strip() != ""]))
    return gpus


def self_launch(args):
    """`python bench.py --gpus N` with no launcher (no WORLD_SIZE in the environment): this process has only parsed its arguments --
    no HIP call, no device query beyond counting -- so it may start the N ranks itself: N fresh children of this interpreter with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (one process per GPU, the same contract torch.distributed.run provides), rank 0's
    stdout relayed as ours, exit code = the worst of the children's.  A rank that dies takes the others down after a grace period
    (they would otherwise wait in the exchange until its time-out)."""
    import socket
    import subprocess
    n = args.gpus
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    have = visible_gpu_count()                            # from the KFD topology: the launcher itself never initialises a HIP runtime
    if have is not None and have < n and env.get("KZG_BENCH_BACKEND", "nccl") == "nccl":
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible.  One rank per GPU over RCCL needs %d devices; to rehearse the N-rank "
                         "path on fewer (every rank on GPU 0, exchange over gloo) set KZG_BENCH_BACKEND=gloo" % (n, have, n))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    if args.test_child_cmd:                               # test hook (tests/test_host_logic.py passes the flag): stand-in ranks, to test the launcher without a GPU
        cmd = args.test_child_cmd.split()
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port), KZG_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen(cmd, env=e, stdout=None if r == 0 else subprocess.DEVNULL))      # rank 0 prints the line; stderr is shared
    import signal

    def stop_ranks(signum, _frame):                       # the launcher is told to stop (driver time-out, Ctrl-C): take the ranks along
        for p in procs:                                   # exactly the PIDs started above
            if p.poll() is None:
                p.terminate()
        raise SystemExit(128 + signum)
    old_handlers = {sig: signal.signal(sig, stop_ranks) for sig in (signal.SIGTERM, signal.SIGINT)}
    worst, failed_at = 0, None
    grace = float(os.environ.get("KZG_BENCH_RANK_GRACE_S", "60"))
    live = list(procs)
    try:
        while live:
            time.sleep(0.2)
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0:
                    worst = worst or (rc if rc > 0 else 128 - rc)
                    failed_at = failed_at or time.monotonic()
            if failed_at and live and time.monotonic() - failed_at > grace:
                for p in live:
                    p.kill()
                failed_at = time.monotonic() + 1e9
    finally:
        for p in procs:                                   # never leave a rank behind on the GPUs
            if p.poll() is None:
                p.kill()
        for sig, h in old_handlers.items():
            signal.signal(sig, h)
    if worst:
        raise SystemExit(worst)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures (profiling runs)")
    ap.add_argument("--test-child-cmd", default=None, help=argparse.SUPPRESS)      # tests only: the command the self-launcher starts per rank
    ap.add_argument("--multi", action="store_true", help="ONE process driving --gpus devices through kzg_multi_* (no torch.distributed, no "
                    "collective): the other partitioning of SURVEY.md 8e; the driver's contract (one rank per GPU over RCCL) is the default mode")
    args, _unknown = ap.parse_known_args()          # (a launcher may append arguments of its own, e.g. --local-rank: ignored)
    if args.multi:
        return main_multi(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)               # no launcher: start the ranks ourselves (nothing has touched the GPU yet)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("KZG_BENCH_FAIL_RANK") == str(rank) and world > 1:
        raise SystemExit(7)                    # test hook (tests/test_gpu_bench_contract.py): a rank that dies before the rendezvous
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d was started by a launcher with WORLD_SIZE=%d: the two must agree" % (args.gpus, world))
    # Rehearsal switch (one-GPU box): KZG_BENCH_BACKEND=gloo runs every rank on GPU 0 and gathers through host memory.
    backend = os.environ.get("KZG_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    # stdout carries ONE line, the JSON: native libraries write to file descriptor 1 themselves (RCCL prints a five-line version banner
    # there when its communicator comes up), so descriptor 1 points at stderr from here on and the line goes out through the saved one.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local_rank)
    # KZG_BENCH_FORCE_EXCHANGE=1: a ONE-rank run takes the N > 1 path end to end -- process group, the per-step all-gather of the partials
    # (RCCL with one rank), fold, every collective below.  What a one-GPU box can rehearse of the driver's multi-GPU run over real RCCL;
    # the line says so ("rccl_ranks": 1, "exchange_backend": "nccl") and its value includes the exchange a single GPU does not need.
    multi = world > 1 or os.environ.get("KZG_BENCH_FORCE_EXCHANGE") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                 # (only the one-rank rehearsal gets here without a launcher's port)
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import rust_kzg_bn254_amd as k
    from rust_kzg_bn254_amd import _lib
    from rust_kzg_bn254_amd.sharding import ShardedMsm

    lib = _lib.load()
    ctx = k.Context(local_rank)
    n = 1 << LOG_N
    tau = int.from_bytes(hashlib.sha256(b"kzg-bn254-mi355x/srs/v1").digest(), "big") % FR

    # ---- inputs: this rank's shard of the SRS (resident) and of the scalars (resident) ---------------------------
    sh = ShardedMsm(ctx, n, rank, world, gather_device="cuda" if backend == "nccl" else None, force_exchange=multi)
    srs = k.SRS.generate(tau, sh.len, ctx=ctx, first_power=sh.lo)
    seed0 = 0x4B5A472D424E3235 & 0x7FFFFFFF
    # Scalars-A, identical on every rank (seeded).  The timed steps ROTATE through N_BUFFERS distinct resident buffers (a commitment
    # service sees fresh scalars; one 32 MiB buffer fed to every step could sit in the 256 MiB Infinity Cache): four independently
    # drawn sets and the same four rotated by half their length (other data order, other addresses; 256 MiB in all at 2^20).
    canon_sets, wire_sets = [], []
    for j in range(N_BUFFERS // 2):
        c = blob_like_canonical(n, seed0 if j == 0 else seed0 + 1 + j)
        canon_sets.append(c); wire_sets.append(ints_to_wire(c))
    for j in range(N_BUFFERS // 2):
        sft = n // 2 + j
        canon_sets.append(canon_sets[j][-sft:] + canon_sets[j][:-sft]); wire_sets.append(np.roll(wire_sets[j], sft, axis=0))
    canon_a, scalars = canon_sets[0], wire_sets[0]
    canon_b, scalars_b = uniform_scalars(n, seed0 + 1)                  # Scalars-B (seed + 1)
    d_sets = [torch.from_numpy(np.ascontiguousarray(w[sh.lo:sh.hi]).view(np.int64)).cuda() for w in wire_sets]
    d_scalars = d_sets[0]
    d_scalars_b = torch.from_numpy(scalars_b[sh.lo:sh.hi].view(np.int64)).cuda()
    rot_ptrs = [d.data_ptr() for d in d_sets]
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # A step = one 2^LOG_N-pair commitment.  Steps are software-pipelined two deep (KZG_BENCH_PIPELINE=0 turns it off):
    # MSM k+1 is enqueued before MSM k is waited for, so every step is still computed and folded inside the timed region.
    pipelined = os.environ.get("KZG_BENCH_PIPELINE", "1") != "0"
    depth_used = (DEPTH or (2 if sh.len >= (1 << 20) else 3)) if pipelined else 1

    # shard-sized steps (N > 1: 2^20 / N pairs per rank) share launches: `GROUP` consecutive steps are ONE batched launch of the MSM engine
    # (ShardedMsm.auto_group: 4 below 2^18 pairs per rank, 2 below 2^20, else 1; KZG_SHARD_GROUP_AUTO=0: off); every step's commitment is still
    # computed, exchanged, folded and checked.  2^17 pairs per rank: 0.164 against 0.22-0.24 ms per step in steady state, 0.19-0.20 against 0.25
    # in a 20-step run
    GROUP = sh.auto_group(srs) if pipelined else 1
    if multi:                                                           # every rank must batch alike (the exchanges are collectives)
        g = torch.tensor([GROUP], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(g, op=dist.ReduceOp.MIN)
        GROUP = int(g.item())
    trace = os.environ.get("KZG_BENCH_TRACE") == "1"                    # per-step completion times on stderr (diagnostics)

    def run_steps(count, ptr, depth, bucket=None, keep=None):
        """`ptr`: one device pointer, or a list the steps cycle through.  keep: list receiving every step's result."""
        res = None
        t_prev = time.perf_counter()
        marks = []
        ptrs = [ptr[i % len(ptr)] for i in range(count)] if isinstance(ptr, list) else [ptr] * count
        if bucket is None and os.environ.get("KZG_BENCH_EXCHANGE_BUCKET"):          # diagnostics: partials per exchange (default 8)
            bucket = int(os.environ["KZG_BENCH_EXCHANGE_BUCKET"])
        for res in sh.commit_stream(srs, ptrs, depth=depth, bucket=bucket, group=(GROUP if depth > 1 else 1)):
            if keep is not None:
                keep.append(res)
            if trace:
                t = time.perf_counter(); marks.append((t - t_prev) * 1e3); t_prev = t
        if trace and rank == 0:
            print("steps(%d, depth %d) ms: %s" % (count, depth, " ".join("%.2f" % m for m in marks)), file=sys.stderr, flush=True)
        return res

    rank_elapsed = []                                                   # every rank's wall time of the last timed() region

    def timed(count, ptr, depth, bucket=None, keep=None):
        barrier()
        t0 = time.perf_counter()
        res = run_steps(count, ptr, depth, bucket, keep)
        barrier()
        el = time.perf_counter() - t0
        if multi:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            rank_elapsed[:] = [float(v.item()) for v in every]
            el = max(rank_elapsed)                                         # MAX over ranks
        else:
            rank_elapsed[:] = [el]
        return el, res

    run_steps(depth_used, d_scalars.data_ptr(), depth_used)             # set-up: every slot allocates its workspace once
    # set-up, continued: bring the process and the GPU to the state a commitment service runs in (KZG_BENCH_TRACE=1 prints the
    # per-step times this is based on).  (1) The first work submitted after the first device-wide synchronisation of the process
    # stalls once for 4-5 ms (runtime-side, seen as a 6.0 ms first step): with the barrier of timed() as that first synchronisation
    # the stall sat INSIDE the timed region -- 0.23 ms per step at --steps 20.  (2) The host spent seconds preparing the inputs
    # above with the GPU idle; the clock needs ~30 steps to settle (per-step times 1.45 -> 1.15 ms).
    # The same step count on every rank (the exchange is a collective); none of this is timed, the W warm-up steps follow.
    # (3) Python's cyclic collector is switched off from here to the end of the measurements (KZG_BENCH_GC=1 leaves it on): with torch
    # imported a full collection takes ~5 ms on the host thread that feeds the GPU, and the container allocations of the exchange path
    # (N > 1) trigger one ~40 steps into a region -- the GPU drains (depth 2: 1.1 ms of queued work), idles 4 ms and then spends ~15
    # steps regaining its clock: 5.6-5.9 ms lost once per region, = 30 shard steps at 8 ranks (profiles/r04_exchange_gc_stall.txt).
    # A service would gc.freeze() after set-up for the same reason.  Nothing of the measured path runs in the collector.
    import gc
    gc_off = os.environ.get("KZG_BENCH_GC", "0") == "0"
    if gc_off:
        gc.collect()
        gc.disable()
    barrier()
    spinup_steps = int(os.environ.get("KZG_BENCH_SPINUP_STEPS", str(min(48 * world, 384))))
    # what a COLD service sees: the first 20 steps after the idle set-up, before the clock has settled -- reported, never `value`
    cold_n = min(20, spinup_steps)
    torch.cuda.synchronize()
    t_cold = time.perf_counter()
    run_steps(cold_n, rot_ptrs, depth_used)
    torch.cuda.synchronize()
    cold_ms_per_step = (time.perf_counter() - t_cold) / max(1, cold_n) * 1e3
    clocks = ClockSampler(local_rank)                                  # engine clock under this very load, sampled in the UNTIMED spin-up only:
    with clocks:                                                        # inside a timed region the sysfs reads cost 4-10 % (1.15 -> 1.27 ms per step at 20 steps)
        run_steps(spinup_steps - cold_n, rot_ptrs, depth_used)
    run_steps(args.warmup, rot_ptrs, depth_used)                        # the W untimed warm-up steps
    timed_results = []
    elapsed, result = timed(args.steps, rot_ptrs, depth_used, keep=timed_results)    # THE timed region: exactly --steps steps, step k on buffer k mod N_BUFFERS
    per_rank_ms = [e / args.steps * 1e3 for e in rank_elapsed]
    comm_ranks = dist.get_world_size() if multi else 1
    if multi:
        # every rank must hold the same folded commitment
        chk = torch.from_numpy(result.view(np.int64).copy())
        if backend == "nccl":
            chk = chk.cuda()
        ref = chk.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, chk), "ranks disagree on the folded commitment"

    # ---- outside the timed region: the other two headline numbers and the kernel profile -------------------------
    side_steps = max(4, min(args.steps, 20))
    run_steps(2, d_scalars_b.data_ptr(), depth_used)
    elapsed_b, result_b = timed(side_steps, d_scalars_b.data_ptr(), depth_used)      # uniform scalars, same pipelining
    run_steps(2, d_scalars.data_ptr(), 1, 1)
    elapsed_lat, result_lat = timed(side_steps, d_scalars.data_ptr(), 1, 1)          # one commitment at a time, one exchange per step: latency
    # kernel durations: HIP events recorded by the library on its launch streams, one MSM at a time (no other MSM shares the GPU)
    lib.kzg_ctx_set_profiling(ctx.handle, 1)
    run_steps(side_steps, d_scalars.data_ptr(), 1)
    barrier()
    phase = (C.c_double * 8)()
    launches, pairs = C.c_uint64(0), C.c_uint64(0)
    lib.kzg_ctx_get_msm_profile(ctx.handle, phase, C.byref(launches), C.byref(pairs))
    prof_entries = C.c_uint64(0)
    lib.kzg_ctx_get_msm_profile_entries(ctx.handle, C.byref(prof_entries))
    lib.kzg_ctx_set_profiling(ctx.handle, 0)
    phase_alone = [phase[i] / max(1, launches.value) for i in range(8)]
    units_per_launch = pairs.value / max(1, launches.value)
    entries_per_launch = prof_entries.value / max(1, launches.value)      # sorted entries = mixed additions (NAF mode: data dependent)
    lib.kzg_ctx_set_profiling(ctx.handle, 1)
    run_steps(side_steps, d_scalars.data_ptr(), depth_used)
    barrier()
    lib.kzg_ctx_get_msm_profile(ctx.handle, phase, C.byref(launches), C.byref(pairs))
    lib.kzg_ctx_set_profiling(ctx.handle, 0)
    phase_piped = [phase[i] / max(1, launches.value) for i in range(8)]

    # ---- BASELINE config 4 ("Full blob -> commit + proof, 2^20 SRS, sharded MSM") at every world size, outside `value` -------------------------
    # evaluations of one polynomial in HOST memory -> commitment and proof at z = Scalars-B[0], through
    #   "lagrange" : evaluation-index shards of the Lagrange basis (csrc/lagrange.hip): a rank uploads, inverts and divides ITS slice only;
    #                two small exchanges per proof (partial barycentric sums -> y, partial points)              [prover/src/kzg.rs:96-100, :128-178]
    #   "replicated": round 4's kzg_*_partial -- every rank uploads, transforms (IFFT) and divides the WHOLE polynomial, then commits its slice
    # Local failures are reported, never raised, so that every rank still joins the collectives.  KZG_BENCH_CONFIG4=0 skips it.
    config4 = None
    if os.environ.get("KZG_BENCH_CONFIG4", "1") != "0" and not args.no_secondary and LOG_N <= 22:
        from rust_kzg_bn254_amd.sharding import ShardedKzg, ShardedKzgLagrange
        c4_reps = max(2, min(8, args.steps))
        c4 = {"ok": True, "err": None}
        gdev = "cuda" if backend == "nccl" else None
        ev_wire = wire_sets[1]                                           # Scalars-A set 1 as EVALUATIONS (identical on every rank)
        z_c4 = np.ascontiguousarray(scalars_b[0])
        lag_shard = None
        try:
            t_su = time.perf_counter()
            os.environ["KZG_NO_PRECOMPUTE"] = "1"                        # the points only: g1_ifft reads them once
            try:
                srs_plain = k.SRS.generate(tau, n, ctx=ctx)
            finally:
                del os.environ["KZG_NO_PRECOMPUTE"]
            lag_shard = srs_plain.lagrange_shard(n, sh.lo, sh.len)       # KZG::g1_ifft(n) once per rank, its slice kept (with tables)
            srs_plain.close()
            c4["setup_s"] = time.perf_counter() - t_su
            sk_new = ShardedKzgLagrange(ctx, lag_shard, n, rank, world, gather_device=gdev, force_exchange=multi)
            sk_old = ShardedKzg(ctx, srs, n, rank, world, gather_device=gdev)
            poly_c4 = k.PolynomialEvalForm(ev_wire)
        except Exception as e:                                           # noqa: BLE001
            c4["ok"], c4["err"] = False, "%s: %s" % (type(e).__name__, e)

        if multi:                                                        # a rank whose SET-UP failed would skip the calls below, whose collectives its peers then wait in:
            okt = torch.tensor([1 if c4["ok"] else 0], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)                    # every rank runs config 4, or none does
            if not bool(okt.item()) and c4["ok"]:
                c4["ok"], c4["err"] = False, "set-up failed on another rank"

        def c4_region(fn):
            """reps calls of fn between barriers; (seconds per call, MAX over ranks; last result)"""
            res = None
            try:
                if c4["ok"]:
                    fn()                                                 # untimed: workspaces, tables
            except Exception as e:                                       # noqa: BLE001
                c4["ok"], c4["err"] = False, "%s: %s" % (type(e).__name__, e)
            barrier()
            t0_ = time.perf_counter()
            try:
                if c4["ok"]:
                    for _ in range(c4_reps):
                        res = fn()
            except Exception as e:                                       # noqa: BLE001
                c4["ok"], c4["err"] = False, "%s: %s" % (type(e).__name__, e)
            barrier()
            el_ = time.perf_counter() - t0_
            if multi:
                t_ = torch.tensor([el_], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                el_ = float(t_.item())
            return el_ / c4_reps, res

        c4["new_commit_s"], c4["new_commit"] = c4_region(lambda: sk_new.commit_eval_form(ev_wire))
        c4["new_proof_s"], c4["new_proof"] = c4_region(lambda: sk_new.compute_proof(ev_wire, z_c4, want_y=True))
        # the same through the STREAM (sharding.commit_and_prove_stream: commitment and proof of a blob from one upload, two blobs in flight)
        def c4_stream():
            last = None
            for last in sk_new.commit_and_prove_stream([(ev_wire, z_c4)] * c4_reps):
                pass
            return last
        c4["stream_s"], c4["stream_last"] = c4_region(c4_stream)
        c4["stream_s"] /= c4_reps                                        # c4_region divides by its repetitions; each repetition streams c4_reps blobs
        c4["old_commit_s"], c4["old_commit"] = c4_region(lambda: sk_old.commit_eval_form(poly_c4))
        c4["old_proof_s"], c4["old_proof"] = c4_region(lambda: sk_old.compute_proof(poly_c4, z_c4, want_y=True))
        if multi:
            okt = torch.tensor([1 if c4["ok"] else 0], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            c4["ok_everywhere"] = bool(okt.item())
        else:
            c4["ok_everywhere"] = c4["ok"]
        if lag_shard is not None:
            lag_shard.close()
        config4 = c4

    # ---- N = 1 only: what every rank of an N-rank run computes per step, on THIS box ---------------------------------
    # A 2^20-pair commitment over N ranks is 2^20 / N pairs per rank.  Rank-sized steps (2^19 / 2^18 / 2^17 pairs over an SRS shard of that
    # length, grouped launches as ShardedMsm.auto_group picks them for that shard size) run here through the WHOLE N > 1 path: nccl process
    # group (one rank -- RCCL refuses two ranks on one device), bucketed all_gather_into_tensor on the gatherer's stream, fold.  From the
    # step times, the strong-scaling efficiency an N-GPU run of this bench can reach: t(2^20) / (N t(2^20 / N)).  20-step regions (the driver's
    # default K) and 96-step regions (steady state).  KZG_BENCH_REHEARSAL=0 skips it.
    shard_rehearsal = None
    if world == 1 and LOG_N >= 14 and not args.no_secondary and os.environ.get("KZG_BENCH_REHEARSAL", "1") != "0":
        shard_rehearsal = {"is": "per-rank step of a 2^%d-pair commitment split N ways, measured on ONE GPU through the N > 1 code path (grouped launches, "
                                 "one-rank RCCL all-gather, fold); efficiency = t(2^%d pairs) / (N x t(2^%d / N pairs)); not a multi-GPU measurement" % (LOG_N, LOG_N, LOG_N),
                           "exchange": None, "sizes": {}}
        try:
            if not multi:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                if "MASTER_PORT" not in os.environ:
                    import socket
                    with socket.socket() as so:
                        so.bind(("127.0.0.1", 0))
                        os.environ["MASTER_PORT"] = str(so.getsockname()[1])
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
            shard_rehearsal["exchange"] = "nccl (RCCL), 1 rank"

            def region(shm, srs_r, ptrs, count, depth, group):
                torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
                t0_ = time.perf_counter()
                last = None
                for last in shm.commit_stream(srs_r, [ptrs[i % len(ptrs)] for i in range(count)], depth=depth, group=group):
                    pass
                torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
                return (time.perf_counter() - t0_) / count * 1e3, last

            # the 2^20 step itself through the same exchange path (N = 1 reference of the ratios), same box, same minute
            sh_ref = ShardedMsm(ctx, n, 0, 1, gather_device="cuda", force_exchange=True)
            region(sh_ref, srs, rot_ptrs, 24, depth_used, 1)
            ref20, _ = region(sh_ref, srs, rot_ptrs, 20, depth_used, 1)
            ref96, res_ref = region(sh_ref, srs, rot_ptrs, 96, depth_used, 1)
            shard_rehearsal["step_2_%d_ms" % LOG_N] = {"20_steps": ref20, "96_steps": ref96}
            ok_all = bool(np.array_equal(res_ref, expected_commitment(canon_sets[95 % N_BUFFERS], tau)))
            for lg, ranks in ((LOG_N - 1, 2), (LOG_N - 2, 4), (LOG_N - 3, 8)):
                per = 1 << lg
                srs_r = k.SRS.generate(tau, per, ctx=ctx)                  # the shard of rank 0: powers [0, per)
                shm = ShardedMsm(ctx, per, 0, 1, gather_device="cuda", force_exchange=True)
                ptrs = [d.data_ptr() for d in d_sets]                       # the first `per` scalars of every resident buffer
                grp = shm.auto_group(srs_r)
                dep = shm.group_depth(3, grp)
                region(shm, srs_r, ptrs, 48, dep, grp)
                t20, _ = region(shm, srs_r, ptrs, 20, dep, grp)
                t96, last = region(shm, srs_r, ptrs, 96, dep, grp)
                want_r = expected_commitment(canon_sets[95 % N_BUFFERS][:per], tau)
                exact_r = bool(np.array_equal(last, want_r))
                ok_all = ok_all and exact_r
                shard_rehearsal["sizes"]["2^%d" % lg] = {"ranks": ranks, "ms_per_step_20": t20, "ms_per_step_96": t96, "steps_per_launch": grp, "pipeline_depth": dep,
                                                        "efficiency_20": ref20 / (ranks * t20), "efficiency_96": ref96 / (ranks * t96), "bit_exact_vs_oracle": exact_r}
                srs_r.close()
            shard_rehearsal["bit_exact_vs_oracle"] = ok_all
            # the same for BASELINE config 4: the per-rank stream of (commitment, proof) over a 2^LOG_N / N-element slice of the evaluations and of
            # the Lagrange basis, resident slices, two blobs in flight, both exchanges through the one-rank nccl group.  One rank's rows do not
            # fold to the true y / proof (the other ranks' parts are missing): the WORK is what each rank of an N-rank run does, the values are
            # checked where all parts exist (bench `config4`, tests/test_gpu_lagrange_shards.py, tests/test_sharding_gloo.py).
            from rust_kzg_bn254_amd.sharding import ShardedKzgLagrange as _SKL
            os.environ["KZG_NO_PRECOMPUTE"] = "1"
            try:
                plain_r = k.SRS.generate(tau, n, ctx=ctx)
            finally:
                del os.environ["KZG_NO_PRECOMPUTE"]
            c4r = {}
            z_r = np.ascontiguousarray(scalars_b[0])
            for lg, ranks in ((LOG_N, 1), (LOG_N - 1, 2), (LOG_N - 2, 4), (LOG_N - 3, 8)):
                per = 1 << lg
                lag_r = plain_r.lagrange_shard(n, 0, per)
                skl = _SKL(ctx, lag_r, n, 0, 1, gather_device="cuda", bounds=(0, per), force_exchange=True)
                items = [(d_sets[i % N_BUFFERS].data_ptr(), z_r) for i in range(24)]
                list(skl.commit_and_prove_stream(items[:6], resident=True))
                torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
                t0_ = time.perf_counter()
                cnt = sum(1 for _ in skl.commit_and_prove_stream(items, resident=True))
                torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
                c4r["2^%d" % lg] = {"ranks": ranks, "ms_per_blob": (time.perf_counter() - t0_) / cnt * 1e3, "blobs": cnt}
                lag_r.close()
            plain_r.close()
            base = c4r["2^%d" % LOG_N]["ms_per_blob"]
            for v in c4r.values():
                v["efficiency"] = base / (v["ranks"] * v["ms_per_blob"])
            shard_rehearsal["config4_stream"] = {"is": "commitment + proof per blob, resident evaluation slices, two blobs in flight, exchanges of 64 B and 384 B per blob "
                                                       "through the one-rank nccl group; efficiency = t(1 rank) / (N x t(slice of 1 / N))", "sizes": c4r}
            if not multi:
                dist.destroy_process_group()
        except Exception as e:                                              # noqa: BLE001 -- never takes the headline down
            shard_rehearsal["error"] = "%s: %s" % (type(e).__name__, e)

    # ---- N > 1 only, outside `value`: the OTHER way to spend N GPUs on a stream of commitments -- whole commitments per rank, no exchange
    # ("replicas": SURVEY 8e's throughput mode).  north_star prescribes the sharded form above (one commitment split N ways: lower latency
    # per commitment, 82-97 % of linear); replicas run every GPU at the one-GPU rate.  Reported so that a user can choose; KZG_BENCH_REPLICAS=0
    # skips it.  Local failures are reported, never raised: every rank still joins the collectives below.
    replicas = None
    if world > 1 and os.environ.get("KZG_BENCH_REPLICAS", "1") != "0":
        rep_ok, rep_err, rep_res, rep_count = True, None, None, 2 * side_steps
        srs_full = None
        try:
            srs_full = k.SRS.generate(tau, n, ctx=ctx)
            d_full = [torch.from_numpy(np.ascontiguousarray(w).view(np.int64)).cuda() for w in wire_sets[:2]]
            shr = ShardedMsm(ctx, n, 0, 1, gather_device=None, force_exchange=False)
            full_ptrs = [d.data_ptr() for d in d_full]

            def run_full(count):
                last = None
                for last in shr.commit_stream(srs_full, [full_ptrs[i % 2] for i in range(count)], depth=2, group=1):
                    pass
                return last
            run_full(24)
        except Exception as e:                                       # noqa: BLE001
            rep_ok, rep_err = False, "%s: %s" % (type(e).__name__, e)
        barrier()
        t0 = time.perf_counter()
        try:
            if rep_ok:
                rep_res = run_full(rep_count)
        except Exception as e:                                       # noqa: BLE001
            rep_ok, rep_err = False, "%s: %s" % (type(e).__name__, e)
        barrier()
        rep_el = time.perf_counter() - t0
        dev = "cuda" if backend == "nccl" else "cpu"
        t_el = torch.tensor([rep_el], dtype=torch.float64, device=dev); dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        t_ok = torch.tensor([1 if rep_ok else 0], dtype=torch.int64, device=dev); dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
        replicas = {"ok_on_every_rank": bool(t_ok.item()), "error_on_rank_0": rep_err, "commitments_per_rank": rep_count,
                    "elapsed_s_max_over_ranks": float(t_el.item()), "result": rep_res, "buffer": (rep_count - 1) % 2}
        if srs_full is not None:
            srs_full.close()
    if gc_off:
        gc.enable()

    exit_code = 0
    if rank == 0:
        # parity at EVERY world size: the folded commitment against big-integer arithmetic on the inputs
        wants = [expected_commitment(c, tau) for c in canon_sets]
        want_a = wants[0]
        want_b = expected_commitment(canon_b, tau)
        exact = bool(np.array_equal(result_lat, want_a) and np.array_equal(result_b, want_b) and len(timed_results) == args.steps
                     and all(np.array_equal(r, wants[i % N_BUFFERS]) for i, r in enumerate(timed_results)))
        if not exact:
            exit_code = 3
        replicas_out = None
        if replicas is not None:
            rep_exact = replicas["result"] is not None and bool(np.array_equal(replicas["result"], wants[replicas["buffer"]]))
            if replicas["ok_on_every_rank"] and not rep_exact:
                exit_code = 3
            rate = world * replicas["commitments_per_rank"] / replicas["elapsed_s_max_over_ranks"] if replicas["ok_on_every_rank"] else None
            replicas_out = {"is": "NOT `value`: every rank commits whole 2^%d-pair polynomials over its own copy of the SRS (no exchange); "
                                  "the sharded form above is the one north_star prescribes" % LOG_N,
                            "commitments_per_s": rate, "pairs_per_s": rate * n if rate else None,
                            "ms_per_commitment_per_gpu": (replicas["elapsed_s_max_over_ranks"] / replicas["commitments_per_rank"] * 1e3) if rate else None,
                            "commitments_per_rank": replicas["commitments_per_rank"], "bit_exact_vs_oracle": rep_exact if replicas["ok_on_every_rank"] else None,
                            "error_on_rank_0": replicas["error_on_rank_0"]}
        config4_out = None
        if config4 is not None:
            config4_out = {"is": "BASELINE config 4, NOT `value`: one evaluation-form polynomial of 2^%d elements in HOST memory -> commitment and proof at "
                                 "z = Scalars-B[0], sharded over %d rank(s); seconds are per call, MAX over ranks, %d calls between barriers" % (LOG_N, world, c4_reps),
                           "ok_on_every_rank": config4.get("ok_everywhere"), "error_on_rank_0": config4.get("err"), "lagrange_shard_setup_s": config4.get("setup_s")}
            if config4.get("ok_everywhere"):
                ev_canon = canon_sets[1]
                z_int = canon_b[0]
                ftau, fz = eval_form_ground_truth(ev_canon, LOG_N, [tau, z_int])
                import pyref
                want_c4 = pyref.point_to_wire(pyref.ec_mul(ftau, (1, 2)))
                want_p4 = pyref.point_to_wire(pyref.ec_mul((ftau - fz) * pow(tau - z_int, -1, FR) % FR, (1, 2)))
                want_y4 = ints_to_wire([fz])[0]
                new_ok = bool(np.array_equal(config4["new_commit"], want_c4) and np.array_equal(config4["new_proof"][0], want_p4) and np.array_equal(config4["new_proof"][1], want_y4))
                st_last = config4["stream_last"]
                stream_ok = bool(st_last is not None and np.array_equal(st_last[0], want_c4) and np.array_equal(st_last[1], want_p4) and np.array_equal(st_last[2], want_y4))
                if not stream_ok:
                    exit_code = 3
                old_ok = bool(np.array_equal(config4["old_commit"], want_c4) and np.array_equal(config4["old_proof"][0], want_p4) and np.array_equal(config4["old_proof"][1], want_y4))
                if not (new_ok and old_ok):
                    exit_code = 3
                per_rank_mib = 32.0 * n / world / 2 ** 20
                config4_out.update({
                    "lagrange_shards": {"commit_ms": config4["new_commit_s"] * 1e3, "proof_ms": config4["new_proof_s"] * 1e3,
                                        "commit_plus_proof_per_s": 1.0 / (config4["new_commit_s"] + config4["new_proof_s"]),
                                        "per_rank": "uploads %.1f MiB, inverts and divides 2^%d / %d elements, MSM over its slice; exchanges: 128 B (commitment), 64 B + 256 B (proof) per rank" % (per_rank_mib, LOG_N, world),
                                        "bit_exact_vs_big_integers": new_ok},
                    "lagrange_shards_streamed": {"commit_plus_proof_ms": config4["stream_s"] * 1e3, "commit_plus_proof_per_s": 1.0 / config4["stream_s"],
                                                 "is": "commit_and_prove_stream: one upload per blob for both results, two or three blobs in flight per rank (grouped launches where the slice allows), "
                                                       "%d blobs per timed stream, %d streams between barriers" % (c4_reps, c4_reps),
                                                 "bit_exact_vs_big_integers": stream_ok},
                    "replicated_r4": {"commit_ms": config4["old_commit_s"] * 1e3, "proof_ms": config4["old_proof_s"] * 1e3,
                                      "commit_plus_proof_per_s": 1.0 / (config4["old_commit_s"] + config4["old_proof_s"]),
                                      "per_rank": "uploads %.0f MiB, IFFT and quotient of all 2^%d elements on every rank, MSM over its slice; exchange: 128 B per rank" % (32.0 * n / 2 ** 20, LOG_N),
                                      "bit_exact_vs_big_integers": old_ok},
                    "bit_exact_check": "commitment == f^(tau) G1, y == f^(z), proof == ((f^(tau) - y) / (tau - z)) G1 with f^ by big-integer barycentric evaluation (bench.py eval_form_ground_truth)"})
        ms_per_step = elapsed / args.steps * 1e3
        pairs_per_s = n * args.steps / elapsed
        acc_ms = phase_alone[4]                                   # k_msm_accumulate, average launch duration, running alone
        achieved = BYTES_PER_PAIR * units_per_launch / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic_bytes(LOG_N) if world == 1 else (None, "PMC passes are single-GPU")
        plan = {"entries": entries_per_launch} if entries_per_launch > 0 else None
        # table mode (two-level sort from the scalars): coarse histogram | scan + pass 1 | pass 2; other modes: digits | histogram + scan | scatter
        phase_names = ["digits_or_coarse_hist", "sort_pass1", "sort_pass2", "unused", "accumulate", "bucket_sums_reduce1", "reduce2", "device_total"]
        out = {
            "metric": "G1-MSM (scalar,point) pairs/s = 2^%d x KZG coeff-form commitments/s, 2^%d-point SRS" % (LOG_N, LOG_N),
            "value": pairs_per_s,
            "unit": "pairs/s",
            "commitments_per_s": args.steps / elapsed,
            "value_uniform": n * side_steps / elapsed_b,
            "latency_ms": elapsed_lat / side_steps * 1e3,
            "n_gpus": world,
            "rccl_ranks": comm_ranks, "exchange_backend": (backend if multi else None),
            "ms_per_step_per_rank": per_rank_ms,
            "launched_by": "bench.py itself (N child ranks)" if os.environ.get("KZG_BENCH_SELF_LAUNCHED") else ("external launcher" if world > 1 else "single process"),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32x9 (29-bit limbs, 64-bit accumulate)",
            "data": "synthetic: SRS P_i = tau^i G1 with known tau (generated on device); value: blob-like scalars < 2^248 (Scalars-A), "
                    "value_uniform: uniform scalars in [0, r) (Scalars-B, seed + 1); seeded",
            "config": {"workload": "G1 MSM 2^%d scalars (KZG::commit_coeff_form), scalars resident in HBM; step k commits buffer k mod %d "
                                   "(%d distinct resident scalar sets, %d MiB)" % (LOG_N, N_BUFFERS, N_BUFFERS, N_BUFFERS * 32 * n >> 20),
                       "sharding": "by scalar index over %d GPU(s); all-gather of XYZZ partials + host fold" % world,
                       "host_gc_in_timed_region": os.environ.get("KZG_BENCH_GC", "0") != "0",
                       "pipeline_depth": sh.group_depth(depth_used, GROUP), "steps_per_launch": GROUP,
                       "cold_start_ms_per_step": cold_ms_per_step,
                       "cold_start_is": "the first %d steps after the idle set-up (engine clock not yet settled), this rank; untimed, reported only" % cold_n,
                       "untimed_before_warmup": "set-up: %d steps (workspaces), the first device-wide synchronisation, %d steps (clock ramp); then the %d warm-up steps" % (depth_used, spinup_steps, args.warmup),
                       "latency_ms_is": "one commitment at a time (depth 1, one exchange per step), %d steps" % side_steps,
                       "bit_exact_vs_oracle": exact,
                       "bit_exact_check": "EVERY timed step's folded commitment == (sum_i c_i tau^i mod r) * G1 of its buffer by big-integer arithmetic "
                                          "+ one affine scalar multiplication (tests/pyref.py); likewise value_uniform and latency_ms; on rank 0 at every world size"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_msm_accumulate", "avg_launch_ms": acc_ms,
                         "avg_launch_ms_pipelined": phase_piped[4],
                         "algorithmic_bytes_per_launch": BYTES_PER_PAIR * units_per_launch,
                         "note": "the binding resource is integer-VALU issue (254-bit modular multiply), see `valu`; traffic (PMC) exceeds the "
                                 "algorithmic bytes by design: one 64-byte precomputed-table point is gathered per (scalar, digit)"},
            "gpu_clock_under_load": clocks.report(),
            "replicas_mode": replicas_out,
            "config4": config4_out,
            "shard_rehearsal": shard_rehearsal,
            "phases_ms_per_launch": dict(zip(phase_names, phase_alone)),
            "phases_ms_per_launch_pipelined": dict(zip(phase_names, phase_piped)),
        }
        if plan and acc_ms > 0:
            # VALU roofline of the same kernel: multiply-adds it must issue / the v_mad_i64_i32 issue rate of THIS chip, measured in this run
            rates = (C.c_double * 6)()
            measured = world == 1 and lib.kzg_ctx_measure_valu_rates(ctx.handle, 3, rates) == 0
            rates = tuple(rates[i] for i in range(6)) if measured else VALU_RATES_FALLBACK_NS
            entries = plan["entries"]
            floor_ms = entries / 64.0 * MADS_PER_MIXED_ADD * rates[0] / N_SIMDS * 1e-6
            mix_ns = sum(c * r for c, r in zip(VALU_MIX_COUNTS, rates))
            mix_floor_ms = entries / 64.0 * mix_ns / N_SIMDS * 1e-6
            out["roofline"]["valu"] = {"mixed_adds_per_launch": entries, "mixed_adds_per_pair": entries / max(1.0, units_per_launch),
                                       "mixed_adds_source": "sorted entries of the profiled launches, counted on the device (per-bit SRS tables, width-18 NAF digits: "
                                                            "~13.8 per scalar; 15 with the fixed 17-bit windows of an SRS uploaded under KZG_NO_NAF=1)",
                                       "mads_per_mixed_add": MADS_PER_MIXED_ADD,
                                       "mad_issue_floor_ms": floor_ms, "frac_of_mad_issue_floor": floor_ms / acc_ms,
                                       "instruction_issue_floor_ms": mix_floor_ms, "frac_of_instruction_issue_floor": mix_floor_ms / acc_ms,
                                       "instructions_per_mixed_add": sum(VALU_MIX_COUNTS),
                                       "instruction_issue_floor_is": "all %d instructions of one mixed addition (v_mad_i64_i32, v_mul_lo_u32, v_ashrrev_i64, v_and_b32, "
                                                                     "other, s_nop: %s) at the issue rates below: what this instruction stream costs on a saturated SIMD"
                                                                     % (sum(VALU_MIX_COUNTS), "/".join(str(c) for c in VALU_MIX_COUNTS)),
                                       "rates_ns_per_wave_instruction_per_simd": list(rates),
                                       "rates_source": "measured in this run on this device, 3 waves per SIMD (kzg_ctx_measure_valu_rates)" if measured
                                                       else "profiles/r01_valu_rates_mi355x.txt (another box; world > 1)",
                                       "peak": "v_mad_i64_i32: %.2f ns per wave-instruction per SIMD, %d SIMDs" % (rates[0], N_SIMDS)}
        if world == 1 and not args.no_secondary:
            # secondary figures of the same run (outside the timed region; BASELINE configs 3 and 4 on one GPU)
            def avg_ms(fn, reps=10, warm=2):
                for _ in range(warm):
                    fn()
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t) / reps * 1e3
            def stats_ms(fn, reps=50, warm=5, sched=False):
                """Per-call wall times of a SYNCHRONOUS library call: median (the figure reported), mean, p99, min, max.  A mean over a
                few calls is owned by a single stall (0.214 ms at 512 coefficients against a 0.068 ms median)."""
                for _ in range(warm):
                    fn()
                ts = []; thr = []; rq = []
                for _ in range(reps):
                    w0 = runqueue_wait_ns_by_thread() if sched else {}
                    th0 = cgroup_nr_throttled()
                    t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t) * 1e3)
                    thr.append(th0 is not None and cgroup_nr_throttled() > th0)
                    if sched:
                        w1 = runqueue_wait_ns_by_thread()
                        rq.append(max([(w1[k_] - w0[k_]) / 1e6 for k_ in w1 if k_ in w0] or [float("nan")]))
                slowest = max(range(len(ts)), key=ts.__getitem__)
                slowest_throttled = thr[slowest]
                slowest_rq = rq[slowest] if rq else None
                rq_median = sorted(rq)[len(rq) // 2] if rq else None
                ts.sort()
                out = {"median": ts[len(ts) // 2], "mean": sum(ts) / len(ts), "p99": ts[min(len(ts) - 1, int(0.99 * len(ts)))], "min": ts[0], "max": ts[-1], "calls": len(ts)}
                if cgroup_nr_throttled() is not None:
                    # CFS bandwidth control: calls during which the cgroup's nr_throttled moved, and whether the slowest call was one of them
                    out["calls_in_a_throttled_period"] = sum(thr); out["slowest_call_throttled"] = bool(slowest_throttled)
                if slowest_rq is not None and slowest_rq == slowest_rq:
                    # the longest any ONE thread of the process sat runnable without a CPU during the slowest call (and the median over the calls): the host's scheduler, not the library
                    out["slowest_call_longest_runqueue_wait_ms"] = slowest_rq; out["median_call_longest_runqueue_wait_ms"] = rq_median
                return out
            d_ntt = d_scalars.clone()
            ntt_ms = avg_ms(lambda: lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d_ntt.data_ptr()), n, 0))
            intt_ms = avg_ms(lambda: lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d_ntt.data_ptr()), n, 1))
            o8 = np.zeros(8, np.uint64); o4 = np.zeros(4, np.uint64); oi = C.c_uint8(0)
            zq = np.ascontiguousarray(scalars[12345 % n])
            ce_ms = avg_ms(lambda: lib.kzg_commit_eval_form(ctx.handle, srs.handle, _lib.ptr(scalars), n, _lib.ptr(o8), C.byref(oi)), reps=5)
            pr_ms = avg_ms(lambda: lib.kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(scalars), n, None, n, _lib.ptr(zq), _lib.ptr(o8),
                                                         C.byref(oi), _lib.ptr(o4)), reps=5)
            cc_ms = avg_ms(lambda: lib.kzg_commit_coeff_form(ctx.handle, srs.handle, _lib.ptr(scalars), n, _lib.ptr(o8), C.byref(oi)), reps=5)
            # stream of host-buffer commitments, THREE in flight (kzg_msm_g1_srs_begin / _end): the 32 MiB H2D copy of commitment k + 2
            # runs beside the kernels of k and k + 1 -- the PCIe-inclusive rate of the boundary (1.14-1.16 ms per 2^20 commitment from
            # pageable and pinned caller buffers alike; 1.30 with two in flight: tools/archive/probe_pinned_stream.py)
            def stream_host(reps=8, depth=3):
                inflight = []
                for i in range(reps):
                    if len(inflight) == depth:
                        assert lib.kzg_msm_g1_srs_end(ctx.handle, inflight.pop(0), _lib.ptr(o8), C.byref(oi), None) == 0
                    rc = lib.kzg_msm_g1_srs_begin(ctx.handle, srs.handle, 0, _lib.ptr(scalars), n, i % depth)
                    assert rc == 0, rc
                    inflight.append(i % depth)
                while inflight:
                    assert lib.kzg_msm_g1_srs_end(ctx.handle, inflight.pop(0), _lib.ptr(o8), C.byref(oi), None) == 0
            stream_host(6)
            t = time.perf_counter(); stream_host(24); cc_stream_ms = (time.perf_counter() - t) / 24 * 1e3
            assert np.array_equal(o8, want_a), "streamed host-buffer commitment differs"
            # config 4 front end: blob bytes (host) -> Fr -> INTT -> MSM (kzg_commit_blob), 2^20 elements = 32 MiB of padded bytes
            blob_bytes = np.frombuffer(b"".join(b"\x00" + bytes(r) for r in np.random.default_rng(7).integers(32, 127, size=(n, 31), dtype=np.uint8)), dtype=np.uint8).copy()
            u8p = C.POINTER(C.c_uint8)
            cb_ms = avg_ms(lambda: lib.kzg_commit_blob(ctx.handle, srs.handle, blob_bytes.ctypes.data_as(u8p), blob_bytes.size, _lib.ptr(o8), C.byref(oi)), reps=5)
            def stream_blob(reps=8):
                prev = None
                for i in range(reps):
                    assert lib.kzg_commit_blob_begin(ctx.handle, srs.handle, blob_bytes.ctypes.data_as(u8p), blob_bytes.size, i & 1) == 0
                    if prev is not None:
                        assert lib.kzg_msm_g1_srs_end(ctx.handle, prev, _lib.ptr(o8), C.byref(oi), None) == 0
                    prev = i & 1
                assert lib.kzg_msm_g1_srs_end(ctx.handle, prev, _lib.ptr(o8), C.byref(oi), None) == 0
            want_blob = o8.copy()
            stream_blob(2)
            t = time.perf_counter(); stream_blob(8); cb_stream_ms = (time.perf_counter() - t) / 8 * 1e3
            assert np.array_equal(o8, want_blob), "streamed blob commitment differs"
            def stream_proof(reps=6, depth=3):
                # THREE proofs in flight, as the commitment stream above: the 32 MiB upload of proof k + 2 runs beside the kernels of k and k + 1
                inflight = []
                for i in range(reps):
                    if len(inflight) == depth:
                        assert lib.kzg_compute_proof_end(ctx.handle, inflight.pop(0), _lib.ptr(o8), C.byref(oi), _lib.ptr(o4)) == 0
                    assert lib.kzg_compute_proof_begin(ctx.handle, srs.handle, _lib.ptr(scalars), n, None, n, _lib.ptr(zq), i % depth) == 0
                    inflight.append(i % depth)
                while inflight:
                    assert lib.kzg_compute_proof_end(ctx.handle, inflight.pop(0), _lib.ptr(o8), C.byref(oi), _lib.ptr(o4)) == 0
            assert lib.kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(scalars), n, None, n, _lib.ptr(zq), _lib.ptr(o8), C.byref(oi), _lib.ptr(o4)) == 0
            want_proof = o8.copy()
            stream_proof(4)
            t = time.perf_counter(); stream_proof(12); pr_stream_ifft_ms = (time.perf_counter() - t) / 12 * 1e3
            assert np.array_equal(o8, want_proof), "streamed proof differs"
            stream_proof(4, depth=2)
            t = time.perf_counter(); stream_proof(12, depth=2); pr_stream_ifft_d2_ms = (time.perf_counter() - t) / 12 * 1e3
            assert np.array_equal(o8, want_proof), "streamed proof differs"
            # KZG::compute_blob_proof end to end (kzg.rs:288-309): bytes in, Fiat-Shamir challenge (SHA-256 over 32 MiB on a host
            # thread) and proof out; and commit + proof of the same blob in one call (the hash runs beside the GPU commitment)
            ob = np.zeros(8, np.uint64); oz = np.zeros(4, np.uint64); oy = np.zeros(4, np.uint64); oc = np.zeros(8, np.uint64); oci = C.c_uint8(0)
            assert lib.kzg_commit_blob(ctx.handle, srs.handle, blob_bytes.ctypes.data_as(u8p), blob_bytes.size, _lib.ptr(oc), C.byref(oci)) == 0
            bp_ms = avg_ms(lambda: lib.kzg_compute_blob_proof(ctx.handle, srs.handle, blob_bytes.ctypes.data_as(u8p), blob_bytes.size, n, _lib.ptr(oc),
                                                              _lib.ptr(ob), C.byref(oi), _lib.ptr(oz), _lib.ptr(oy)), reps=4, warm=1)
            oc2 = np.zeros(8, np.uint64); ob2 = np.zeros(8, np.uint64)
            cp_ms = avg_ms(lambda: lib.kzg_commit_and_prove_blob(ctx.handle, srs.handle, blob_bytes.ctypes.data_as(u8p), blob_bytes.size, n, _lib.ptr(oc2),
                                                                 C.byref(oci), _lib.ptr(ob2), C.byref(oi), _lib.ptr(oz), _lib.ptr(oy)), reps=4, warm=1)
            assert np.array_equal(oc2, oc) and np.array_equal(ob2, ob), "commit + proof in one call differs from the two calls"
            ch_ms = avg_ms(lambda: lib.kzg_compute_challenge(blob_bytes.ctypes.data_as(u8p), blob_bytes.size, _lib.ptr(oc), _lib.ptr(oz)), reps=3, warm=1)
            # The same calls with the Lagrange basis of 2^LOG_N points cached on the device (kzg_srs_cache_lagrange: KZG::g1_ifft once instead of inside
            # every commit_eval_form, prover/src/kzg.rs:96-98): eval-form commitments and proofs are ONE MSM over it -- no IFFT (kzg.rs:98-100, :176-177)
            t = time.perf_counter()
            assert lib.kzg_srs_cache_lagrange(ctx.handle, srs.handle, n) == 0
            lag_cache_s = time.perf_counter() - t
            ce_lag_ms = avg_ms(lambda: lib.kzg_commit_eval_form(ctx.handle, srs.handle, _lib.ptr(scalars), n, _lib.ptr(o8), C.byref(oi)), reps=5)
            pr_lag_ms = avg_ms(lambda: lib.kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(scalars), n, None, n, _lib.ptr(zq), _lib.ptr(o8), C.byref(oi), _lib.ptr(o4)), reps=5)
            assert np.array_equal(o8, want_proof), "proof over the cached Lagrange basis differs from the IFFT path"
            stream_proof(4)
            t = time.perf_counter(); stream_proof(12); pr_stream_ms = (time.perf_counter() - t) / 12 * 1e3
            assert np.array_equal(o8, want_proof), "streamed proof over the cached Lagrange basis differs"
            cb_lag_ms = avg_ms(lambda: lib.kzg_commit_blob(ctx.handle, srs.handle, blob_bytes.ctypes.data_as(u8p), blob_bytes.size, _lib.ptr(o8), C.byref(oi)), reps=5)
            assert np.array_equal(o8, want_blob), "blob commitment over the cached Lagrange basis differs"
            # BASELINE config 4 read literally, as a STREAM (kzg_commit_and_prove_blob_begin / _end, csrc/blobstream.hip): blob bytes in host memory ->
            # commitment + Fiat-Shamir challenge + proof, `depth` blobs in flight so that their transcript hashes (one sequential SHA-256 stream of
            # 32 MiB each, compute_challenge_host_sha256_ms) run side by side on host threads while the GPU works through commitments and proofs.
            # Two distinct blobs alternate; every result is compared with the one-call entry's (itself checked against big integers in tests/).
            blob_b = np.ascontiguousarray(np.roll(blob_bytes.reshape(-1, 32), 12345, axis=0).reshape(-1))
            want_cp = []
            for bb in (blob_bytes, blob_b):
                wc = np.zeros(8, np.uint64); wp = np.zeros(8, np.uint64); wz = np.zeros(4, np.uint64); wy = np.zeros(4, np.uint64)
                assert lib.kzg_commit_and_prove_blob(ctx.handle, srs.handle, bb.ctypes.data_as(u8p), bb.size, n, _lib.ptr(wc), C.byref(oci), _lib.ptr(wp), C.byref(oi),
                                                     _lib.ptr(wz), _lib.ptr(wy)) == 0
                want_cp.append((wc, wp, wz, wy))
            assert np.array_equal(want_cp[0][0], oc2) and np.array_equal(want_cp[0][1], ob2), "commit + proof over the cached Lagrange basis differs from the IFFT path"
            def stream_commit_and_prove(total, depth):
                sc = np.zeros(8, np.uint64); sp = np.zeros(8, np.uint64); sz_ = np.zeros(4, np.uint64); sy = np.zeros(4, np.uint64)
                t0 = time.perf_counter()
                for i in range(total + depth):
                    if i >= depth:
                        assert lib.kzg_commit_and_prove_blob_end(ctx.handle, (i - depth) % depth, _lib.ptr(sc), C.byref(oci), _lib.ptr(sp), C.byref(oi), _lib.ptr(sz_), _lib.ptr(sy)) == 0
                        w = want_cp[(i - depth) & 1]
                        assert np.array_equal(sc, w[0]) and np.array_equal(sp, w[1]) and np.array_equal(sz_, w[2]) and np.array_equal(sy, w[3]), "streamed commitment / proof differs"
                    if i < total:
                        bb = blob_b if i & 1 else blob_bytes
                        assert lib.kzg_commit_and_prove_blob_begin(ctx.handle, srs.handle, bb.ctypes.data_as(u8p), bb.size, n, None, i % depth) == 0
                return (time.perf_counter() - t0) / total * 1e3
            cps = {}
            for depth in (8, 12):
                stream_commit_and_prove(depth, depth)
                cps[depth] = min(stream_commit_and_prove(24, depth) for _ in range(2))
            cp_stream_ms = cps[12]
            assert lib.kzg_srs_drop_lagrange(ctx.handle, srs.handle) == 0
            # config 5 shape: verify_kzg_proof_batch core at n = 4096 (three 4096-point MSMs batched on the GPU + host pairing check)
            nb = 4096
            g1w = np.zeros((nb, 8), dtype=np.uint64)
            assert lib.kzg_srs_download(ctx.handle, srs.handle, 0, nb, _lib.ptr(g1w)) == 0
            fr_sel = np.ascontiguousarray(scalars[:nb])
            okf = C.c_int32(0)
            tau_g2 = np.zeros(16, np.uint64)
            lib.kzg_g2_mul_generator(_lib.ptr(k.fr.fr_from_int(tau)), _lib.ptr(tau_g2))
            g1c = np.ascontiguousarray(g1w)
            bv_ms = avg_ms(lambda: lib.kzg_verify_kzg_proof_batch(ctx.handle, _lib.ptr(g1c), _lib.ptr(fr_sel), _lib.ptr(fr_sel), _lib.ptr(g1c), _lib.ptr(fr_sel),
                                                                 nb, _lib.ptr(tau_g2), C.byref(okf)), reps=3, warm=1)
            o3 = np.zeros((3, 8), np.uint64); i3 = np.zeros(3, np.uint8)
            b3 = np.ascontiguousarray(np.concatenate([g1c, g1c, g1c])); s3 = np.ascontiguousarray(np.concatenate([fr_sel, fr_sel, fr_sel]))
            m3_ms = avg_ms(lambda: lib.kzg_msm_g1_batch(ctx.handle, _lib.ptr(b3), _lib.ptr(s3), nb, 3, _lib.ptr(o3), i3.ctypes.data_as(u8p)), reps=10)
            # config 5 END TO END (verifier/src/batch.rs:16-69 in one C-ABI call): 4096 (blob, commitment, proof) rows -- 256 distinct
            # random blobs of 35 .. 50000 raw bytes (verifier/tests/tests.rs:134-192) with their GPU commitments and blob proofs, each
            # used 16 times -- through point validation, 4096 Fiat-Shamir transcripts (host thread pool), 4096 barycentric
            # evaluations (one batched GPU launch), compute_r_powers, three batched GPU MSMs and the host pairing check
            from rust_kzg_bn254_amd.helpers import pad_payload
            rng5 = np.random.default_rng(5)
            rows5 = []
            for n_raw in rng5.integers(35, 50000, size=256):
                data = pad_payload(rng5.integers(32, 127, size=int(n_raw), dtype=np.uint8).tobytes())
                npad = 1
                while npad < len(data) // 32:
                    npad <<= 1
                buf = np.frombuffer(data, dtype=np.uint8)
                c5 = np.zeros(8, np.uint64); p5 = np.zeros(8, np.uint64); ci5 = C.c_uint8(0); pi5 = C.c_uint8(0)
                assert lib.kzg_commit_and_prove_blob(ctx.handle, srs.handle, buf.ctypes.data_as(u8p), len(data), npad, _lib.ptr(c5), C.byref(ci5),
                                                     _lib.ptr(p5), C.byref(pi5), None, None) == 0
                rows5.append((data, c5, p5))
            sel5 = [rows5[i % 256] for i in range(nb)]
            ptrs5, lens5, _keep5 = _lib.blob_args([r[0] for r in sel5])
            cm5 = np.ascontiguousarray(np.stack([r[1] for r in sel5])); pf5 = np.ascontiguousarray(np.stack([r[2] for r in sel5]))
            ok5 = C.c_int32(0)
            e2e_stats = stats_ms(lambda: lib.kzg_verify_blob_kzg_proof_batch(ctx.handle, ptrs5, lens5, _lib.ptr(cm5), _lib.ptr(pf5), nb, _lib.ptr(tau_g2),
                                                                             C.byref(ok5)), reps=40, warm=3, sched=True)
            e2e_ms = e2e_stats["median"]
            assert ok5.value == 1, "the 4096-row batch did not verify"
            pf5[nb - 1] = pf5[0]
            assert lib.kzg_verify_blob_kzg_proof_batch(ctx.handle, ptrs5, lens5, _lib.ptr(cm5), _lib.ptr(pf5), nb, _lib.ptr(tau_g2), C.byref(ok5)) == 0 and ok5.value == 0
            e2e_bytes = sum(len(r[0]) for r in sel5)
            # the reference's own commit / proof bench shapes (prover/benches/bench_kzg_commit.rs:17-42, bench_kzg_proof.rs:17-58: 10 000 .. 50 000
            # byte blobs = 512 .. 2 048 coefficients) from host buffers against the loaded SRS, one call at a time, and g1_ifft(2048)
            small, small_stats = {}, {}
            large_commit, setup_sample = {}, None
            for nn in (512, 1024, 2048):
                sc_s = np.ascontiguousarray(scalars[:nn]); zq_s = np.ascontiguousarray(scalars_b[77])
                def put(name, st):
                    small[name + "_ms"] = st["median"]; small_stats[name] = st
                put("commit_coeff_%d" % nn, stats_ms(lambda: lib.kzg_commit_coeff_form(ctx.handle, srs.handle, _lib.ptr(sc_s), nn, _lib.ptr(o8), C.byref(oi)), reps=60, warm=8))
                # bench_kzg_proof.rs:17-58 proves at a DOMAIN point (compute_proof_with_known_z_fr_index): z = w^idx; the off-domain figure beside it
                roots_s = np.zeros((nn, 4), np.uint64); n_roots = C.c_size_t(0)
                assert lib.kzg_calculate_roots_of_unity(ctx.handle, nn * 32, _lib.ptr(roots_s), nn, C.byref(n_roots)) == 0 and n_roots.value == nn
                zq_on = np.ascontiguousarray(roots_s[(nn * 3) // 7])
                put("compute_proof_%d" % nn, stats_ms(lambda: lib.kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(sc_s), nn, None, nn, _lib.ptr(zq_on), _lib.ptr(o8),
                                                                                     C.byref(oi), _lib.ptr(o4)), reps=60, warm=8))
                put("compute_proof_off_domain_%d" % nn, stats_ms(lambda: lib.kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(sc_s), nn, None, nn, _lib.ptr(zq_s),
                                                                                                _lib.ptr(o8), C.byref(oi), _lib.ptr(o4)), reps=60, warm=8))
            # the same shapes as ONE batched call over 512 polynomials (resident scalars): microseconds per commitment, checked against a single call
            for nn in (512, 2048):
                cnt_b = min(512, n // nn)
                if cnt_b < 2:
                    continue
                ob = np.zeros((cnt_b, 8), np.uint64)
                bms = avg_ms(lambda: lib.kzg_commit_coeff_form_batch_device(ctx.handle, srs.handle, C.c_void_p(d_scalars.data_ptr()), nn, cnt_b, _lib.ptr(ob), None), reps=5, warm=2)
                assert lib.kzg_msm_g1_srs_device(ctx.handle, srs.handle, 0, C.c_void_p(d_scalars.data_ptr() + (cnt_b - 1) * nn * 32), nn, _lib.ptr(o8), C.byref(oi)) == 0
                assert np.array_equal(o8, ob[cnt_b - 1]), "batched commitment differs from the single call"
                small["commit_coeff_%d_batched_x%d_us_per_commitment" % (nn, cnt_b)] = bms * 1e3 / cnt_b
            lag = np.zeros((2048, 8), np.uint64)
            for nn in (512, 1024, 2048):                   # prover/benches/bench_g1_ifft.rs:28-30 sweeps to 2 048
                if n >= nn:
                    st = stats_ms(lambda: lib.kzg_g1_ifft(ctx.handle, srs.handle, nn, _lib.ptr(lag)), reps=9, warm=2)
                    small["g1_ifft_%d_ms" % nn] = st["median"]; small_stats["g1_ifft_%d" % nn] = st
            # verifier/benches/bench_kzg_verify.rs:18-67: ONE verify_proof of a proof at a domain point (host only: [y]G1, [z]G2, two Miller loops, final exponentiation)
            nn = 512
            sc_s = np.ascontiguousarray(scalars[:nn])
            roots_s = np.zeros((nn, 4), np.uint64); n_roots = C.c_size_t(0)
            assert lib.kzg_calculate_roots_of_unity(ctx.handle, nn * 32, _lib.ptr(roots_s), nn, C.byref(n_roots)) == 0
            z_v = np.ascontiguousarray(roots_s[123]); y_v = np.ascontiguousarray(sc_s[123])
            c_v = np.zeros(8, np.uint64); p_v = np.zeros(8, np.uint64); ok_v = C.c_int32(0)
            assert lib.kzg_commit_eval_form(ctx.handle, srs.handle, _lib.ptr(sc_s), nn, _lib.ptr(c_v), C.byref(oi)) == 0
            assert lib.kzg_compute_proof(ctx.handle, srs.handle, _lib.ptr(sc_s), nn, None, nn, _lib.ptr(z_v), _lib.ptr(p_v), C.byref(oi), None) == 0
            st = stats_ms(lambda: lib.kzg_verify_proof(_lib.ptr(c_v), _lib.ptr(p_v), _lib.ptr(y_v), _lib.ptr(z_v), _lib.ptr(tau_g2), C.byref(ok_v)), reps=60, warm=8)
            assert ok_v.value == 1, "verify_proof rejected a correct proof"
            small["verify_proof_ms"] = st["median"]; small_stats["verify_proof"] = st
            # prover/benches/bench_kzg_commit_large_blobs.rs:17-37: commit_coeff_form of an 8 000 000- / 16 252 000-byte blob = 258 065 / 524 259 blob-like
            # coefficients padded to 2^18 / 2^19, from a host buffer, one call at a time
            for name, n_el in (("commit_8mb", 258065), ("commit_16mb", 524259)):
                n_el = n_el if LOG_N >= 20 else max(1, n_el >> (20 - LOG_N))      # (reduced runs of the contract test: the same shapes, scaled)
                npad = 1
                while npad < n_el:
                    npad <<= 1
                sc_l = np.zeros((npad, 4), np.uint64); sc_l[:n_el] = scalars[:n_el]
                st = stats_ms(lambda: lib.kzg_commit_coeff_form(ctx.handle, srs.handle, _lib.ptr(sc_l), npad, _lib.ptr(o8), C.byref(oi)), reps=30, warm=4)
                small[name + "_ms"] = st["median"]; small_stats[name] = st
                large_commit[name] = (sc_l, o8.copy())
            # prover/benches/bench_kzg_setup.rs:6-15: SRS::new(g1.32mb.point, 268435456, 524288) = 524 288 gnark-compressed points (16 MiB) -> decompressed,
            # validated, resident SRS with its window and per-bit tables.  The file is synthetic: the first 2^19 points of the bench's SRS re-encoded.
            n_setup = 524288 if LOG_N >= 20 else max(64, n >> 1)
            P_ = 21888242871839275222246405745257275088696311157297823662689037894645226208583
            rinv = pow(1 << 256, -1, P_)
            wire = np.zeros((n_setup, 8), np.uint64)
            assert lib.kzg_srs_download(ctx.handle, srs.handle, 0, n_setup, _lib.ptr(wire)) == 0
            wb = wire.tobytes()
            enc = bytearray(32 * n_setup)
            half = (P_ - 1) // 2
            for i in range(n_setup):
                x = int.from_bytes(wb[64 * i:64 * i + 32], "little") * rinv % P_
                y = int.from_bytes(wb[64 * i + 32:64 * i + 64], "little") * rinv % P_
                e = x.to_bytes(32, "big")
                enc[32 * i:32 * i + 32] = e
                enc[32 * i] |= 0xC0 if y > half else 0x80        # helpers.rs:175-226: 0b11 = larger y, 0b10 = smaller y
            enc = np.frombuffer(bytes(enc), dtype=np.uint8)
            def setup_once():
                h = C.c_void_p(); bad = C.c_uint64(0)
                assert lib.kzg_srs_load_compressed_be(ctx.handle, enc.ctypes.data_as(u8p), n_setup, C.byref(h), C.byref(bad)) == 0
                return h
            h0 = setup_once()
            back = np.zeros((n_setup, 8), np.uint64)
            assert lib.kzg_srs_download(ctx.handle, h0, 0, n_setup, _lib.ptr(back)) == 0 and np.array_equal(back, wire), "decompressed SRS differs from the points it was encoded from"
            lib.kzg_srs_free(h0)
            ts_setup = []
            for _ in range(5):
                t = time.perf_counter(); hh = setup_once(); ts_setup.append((time.perf_counter() - t) * 1e3); lib.kzg_srs_free(hh)
            ts_setup.sort()
            small["kzg_setup_%d_ms" % n_setup] = ts_setup[len(ts_setup) // 2]
            small_stats["kzg_setup_%d" % n_setup] = {"median": ts_setup[len(ts_setup) // 2], "mean": sum(ts_setup) / len(ts_setup), "min": ts_setup[0], "max": ts_setup[-1], "calls": len(ts_setup)}
            small["kzg_setup_is"] = ("%d gnark-compressed points (%.1f MiB) in host memory -> decompressed (square roots on the GPU), curve-checked, resident, window + per-bit "
                                     "tables built (%.1f GiB)" % (n_setup, n_setup * 32 / 2.0 ** 20, (255 + 15 + 17) * n_setup * 64 / 2.0 ** 30))
            setup_sample = (enc[:32 * min(4096, n_setup)].tobytes(), n_setup)
            small["statistic"] = "median of the per-call wall times (60 calls per commit / proof shape, 9 per g1_ifft size); mean / p99 / min / max under reference_bench_shapes_stats"
            # sizes beyond the tables (parity: tests/test_gpu_large_sizes.py): the 2^20 step WITHOUT the 15.9 GiB of per-bit
            # tables (KZG_NO_NAF=1: fixed 17-bit windows over the 1 GiB window tables), a commitment over a 2^23-point SRS (window tables
            # only: per-bit tables stop at 2^22 points), and the Fr NTT at 2^25 / 2^26 (three passes, lookup twiddles), all device-resident
            beyond = {}
            if LOG_N == 20:
                os.environ["KZG_NO_NAF"] = "1"
                srs_nonaf = k.SRS.generate(tau, n, ctx=ctx)
                del os.environ["KZG_NO_NAF"]
                sh1 = ShardedMsm(ctx, n, 0, 1, gather_device=None)
                list(sh1.commit_stream(srs_nonaf, [rot_ptrs[i % N_BUFFERS] for i in range(24)], depth=2))
                torch.cuda.synchronize(); t = time.perf_counter()
                res_nn = list(sh1.commit_stream(srs_nonaf, [rot_ptrs[i % N_BUFFERS] for i in range(24)], depth=2))
                torch.cuda.synchronize()
                beyond["step_2_20_without_per_bit_tables_ms"] = (time.perf_counter() - t) / 24 * 1e3
                assert all(np.array_equal(r, wants[i % N_BUFFERS]) for i, r in enumerate(res_nn)), "commitment without per-bit tables differs"
                beyond["step_2_20_with_per_bit_tables_ms"] = ms_per_step
                beyond["srs_memory_GiB"] = {"window_tables_c17": 15 * n * 64 / 2.0 ** 30, "small_msm_tables_c15": 17 * n * 64 / 2.0 ** 30, "per_bit_tables": 255 * n * 64 / 2.0 ** 30}
                srs_nonaf.close()
                n23 = 1 << 23
                srs23 = k.SRS.generate(tau, n23, ctx=ctx)
                d23 = torch.cat([d_sets[i % N_BUFFERS] for i in range(8)])               # 2^23 resident scalars (the eight buffers back to back)
                st23 = stats_ms(lambda: lib.kzg_msm_g1_srs_device(ctx.handle, srs23.handle, 0, C.c_void_p(d23.data_ptr()), n23, _lib.ptr(o8), C.byref(oi)), reps=7, warm=2)
                beyond["commit_2_23_pairs_over_2_23_srs_ms"] = st23["median"]; beyond["commit_2_23_pairs_stats"] = st23
                beyond["commit_2_23_pairs_per_s"] = n23 / (st23["median"] * 1e-3)
                srs23.close(); del d23
                for lg in (25, 26):
                    big_n = torch.zeros((1 << lg, 4), dtype=torch.int64, device="cuda")
                    big_n[:, 0] = torch.arange(1 << lg, device="cuda")
                    stn = stats_ms(lambda: (lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(big_n.data_ptr()), 1 << lg, 0), torch.cuda.synchronize()), reps=5, warm=1)
                    beyond["fr_ntt_2_%d_ms" % lg] = stn["median"]
                    beyond["fr_ntt_2_%d_algorithmic_GBps" % lg] = 64.0 * (1 << lg) / (stn["median"] * 1e-3) / 1e9
                    del big_n
            # measured copy ceiling of this box (device-to-device, 1 GiB): read + write bytes per second
            big = torch.empty(1 << 28, dtype=torch.int32, device="cuda"); big2 = torch.empty_like(big)
            copy_ms = avg_ms(lambda: big2.copy_(big), reps=10)
            copy_gbs = 2.0 * big.numel() * 4 / (copy_ms * 1e-3) / 1e9
            del big, big2
            out["secondary"] = {
                "host_buffers_commit_coeff_streamed_ms": cc_stream_ms,
                "commit_blob_from_host_bytes_ms": cb_ms, "commit_blob_from_host_bytes_streamed_ms": cb_stream_ms,
                "compute_blob_proof_from_host_bytes_ms": bp_ms, "commit_and_prove_blob_from_host_bytes_ms": cp_ms,
                "commit_and_prove_blob_streamed_ms": cp_stream_ms, "commit_and_prove_blob_streamed_by_jobs_in_flight_ms": {str(d): v for d, v in cps.items()},
                "commit_and_prove_blob_streamed_is": "32 MiB blobs in host memory -> commitment, challenge and proof per blob, 12 jobs in flight (kzg_commit_and_prove_blob_begin / _end; 8 in flight beside it): "
                                                     "the transcript hashes of the jobs run side by side on host threads; cached Lagrange basis; every result compared",
                "compute_challenge_host_sha256_ms": ch_ms,
                "batch_verify_4096_core_ms": bv_ms, "batch_verify_4096_three_msms_ms": m3_ms,
                "batch_verify_4096_end_to_end_ms": e2e_ms, "batch_verify_4096_end_to_end_stats": e2e_stats, "batch_verify_4096_end_to_end_blob_MiB": e2e_bytes / 2.0 ** 20,
                "batch_verify_4096_end_to_end_host_threads": host_pool_threads(),
                "reference_bench_shapes": small, "reference_bench_shapes_stats": small_stats, "beyond_the_tables": beyond,
                "measured_d2d_copy_GBps": copy_gbs,
                "fr_ntt_ms": ntt_ms, "fr_intt_ms": intt_ms,
                "fr_ntt_algorithmic_GBps": 64.0 * n / (ntt_ms * 1e-3) / 1e9, "fr_ntt_frac_of_hbm_peak": 64.0 * n / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "host_buffers_commit_coeff_ms": cc_ms, "host_buffers_commit_eval_ms": ce_ms, "host_buffers_compute_proof_ms": pr_ms,
                "host_buffers_compute_proof_streamed_ms": pr_stream_ms,
                "host_buffers_compute_proof_streamed_is": "three proofs in flight over the Lagrange basis cached with the SRS (kzg_srs_cache_lagrange, %.2f s once): no IFFT; "
                                                          "the IFFT + monomial-basis path beside it" % lag_cache_s,
                "host_buffers_compute_proof_streamed_ifft_path_ms": pr_stream_ifft_ms, "host_buffers_compute_proof_streamed_ifft_path_two_in_flight_ms": pr_stream_ifft_d2_ms,
                "cached_lagrange_basis": {"cache_once_s": lag_cache_s, "host_buffers_commit_eval_ms": ce_lag_ms, "host_buffers_compute_proof_ms": pr_lag_ms,
                                          "commit_blob_from_host_bytes_ms": cb_lag_ms, "memory_GiB": (255 + 15 + 17) * n * 64 / 2.0 ** 30},
                "note": "NTT on device-resident data (64 B per element algorithmic); host_buffers_* include the 32 MiB H2D copy of the scalars (PCIe)"}
        if world == 1 and not args.no_cpu_baseline:
            import oracle as orc                                   # checker + reported CPU baseline only
            cores = os.cpu_count() or 1
            g1 = srs.g1
            t1 = time.perf_counter()
            want = orc.msm_pippenger(g1, scalars, threads=cores)
            cpu_s = time.perf_counter() - t1
            if not (np.array_equal(want, result_lat) and np.array_equal(want, want_a)):   # second, independent check: the oracle's Pippenger against the GPU's commitment of buffer 0
                out["config"]["bit_exact_vs_oracle"] = False
                exit_code = 3
            out["cpu_baseline"] = {"value": n / cpu_s, "unit": "pairs/s", "cores": min(cores, 17), "kind": "port",
                                   "sample": "the same 2^%d-pair MSM once: oracle/ C restatement of arkworks' signed-window "
                                             "Pippenger (c=15, one thread per window, 17 windows: 17 of the box's %d hardware threads), %.2f s wall" % (LOG_N, cores, cpu_s)}
            m1 = 1 << 16                                           # single-thread sample (SURVEY.md 8d asks for both)
            t1 = time.perf_counter()
            orc.msm_pippenger(g1[:m1], scalars[:m1], threads=1)
            cpu1_s = time.perf_counter() - t1
            out["cpu_baseline"]["single_thread"] = {"value": m1 / cpu1_s, "unit": "pairs/s", "cores": 1,
                                                    "sample": "2^16-pair MSM, same port, 1 thread, %.2f s wall" % cpu1_s}
            # the oracle port beside the reference's bench_kzg_commit_large_blobs / bench_kzg_setup shapes (same inputs as the GPU figures above)
            if not args.no_secondary:
                shapes_cpu = {}
                for name, (sc_l, got_pt) in large_commit.items():
                    t1 = time.perf_counter()
                    want_l = orc.msm_pippenger(g1[:len(sc_l)], sc_l, threads=cores)
                    shapes_cpu[name + "_ms"] = (time.perf_counter() - t1) * 1e3
                    if not np.array_equal(want_l, got_pt):
                        out["config"]["bit_exact_vs_oracle"] = False
                        exit_code = 3
                if setup_sample is not None:
                    t1 = time.perf_counter()
                    sample_bytes, n_setup_pts = setup_sample
                    for i in range(0, len(sample_bytes), 32):
                        orc.g1_decompress_be(sample_bytes[i:i + 32])
                    per_point = (time.perf_counter() - t1) / (len(sample_bytes) // 32)
                    shapes_cpu["kzg_setup_%d_ms" % n_setup_pts] = per_point * n_setup_pts * 1e3
                    shapes_cpu["kzg_setup_is"] = "%d points decompressed one at a time by the oracle (1 thread, %.1f us per point incl. the ctypes call), scaled to %d" % (
                        len(sample_bytes) // 32, per_point * 1e6, n_setup_pts)
                shapes_cpu["commit_is"] = "oracle/ signed-window Pippenger, one thread per window (17 of %d hardware threads), result compared with the GPU's" % cores
                out["secondary"]["reference_bench_shapes_cpu_port"] = shapes_cpu
            # CPU baseline of the Fr NTT (primitives/src/polynomial.rs:130-140, :241-251): radix-2, every layer chunked over the cores
            t1 = time.perf_counter()
            cores = orc.host_cpus()                                # (the layers are chunked over the CPUs the cgroup's quota really grants: 256 threads under a 16-CPU quota run slower)
            cpu_f = orc.fr_ntt_mt(scalars_b, inverse=False, threads=cores)
            ntt_all_s = time.perf_counter() - t1
            t1 = time.perf_counter()
            orc.fr_ntt(scalars_b[:1 << 18], inverse=False)
            ntt_1_s = time.perf_counter() - t1
            d_chk = torch.from_numpy(scalars_b.view(np.int64)).cuda()
            assert lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d_chk.data_ptr()), n, 0) == 0
            ntt_exact = bool(np.array_equal(d_chk.cpu().numpy().view(np.uint64), cpu_f))
            if not ntt_exact:
                exit_code = 3
            out["cpu_baseline_ntt"] = {"value": n / ntt_all_s, "unit": "elements/s", "cores": cores, "kind": "port",
                                       "sample": "one forward 2^%d NTT: oracle/ radix-2 with the layers chunked over %d threads, %.3f s wall" % (LOG_N, cores, ntt_all_s),
                                       "single_thread": {"value": (1 << 18) / ntt_1_s, "unit": "elements/s", "cores": 1,
                                                         "sample": "one forward 2^18 NTT, 1 thread, %.3f s wall" % ntt_1_s},
                                       "gpu_bit_exact_vs_oracle": ntt_exact}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if multi:
        code = torch.tensor([exit_code], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.broadcast(code, src=0)
        exit_code = int(code.item())
        dist.destroy_process_group()
    if exit_code:
        raise SystemExit("bench.py: the commitment differs from the expected point (exit %d)" % exit_code)


if __name__ == "__main__":
    main()
