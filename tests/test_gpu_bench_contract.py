"""-m gpu: bench.py's output contract (one JSON line with the fields the driver reads), on a reduced problem size."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_emits_the_contract_line():
    env = dict(os.environ, KZG_BENCH_LOG_N="14")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and [ln for ln in res.stdout.splitlines() if ln.strip()] == lines        # ONE line on stdout, nothing else
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - (1 << 14) * 6 / (d["ms_per_step"] * 6 * 1e-3)) / d["value"] < 1e-6
    assert "workload" in d["config"] and d["config"]["bit_exact_vs_oracle"] is True
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["value"] > 0
    # round 2: the three headline numbers, the traffic reason, the VALU view of the roofline, the NTT baseline
    assert d["value_uniform"] > 0 and d["latency_ms"] > 0
    assert r["traffic"] is None and len(r["traffic_source"]) > 10     # PMC passes exist for the 2^20 workload only: null + the reason
    assert d["cpu_baseline_ntt"]["value"] > 0 and d["cpu_baseline_ntt"]["gpu_bit_exact_vs_oracle"] is True
    assert "phases_ms_per_launch" in d and "phases_ms_per_launch_pipelined" in d
    # round 5: BASELINE config 4 in the line (both sharding schemes, checked against big integers), the one-GPU rehearsal of the N-rank
    # steps through a real one-rank RCCL exchange with the implied strong-scaling efficiency, proofs over the cached Lagrange basis
    c4 = d["config4"]
    assert c4["ok_on_every_rank"] is True and c4["error_on_rank_0"] is None
    for scheme in ("lagrange_shards", "replicated_r4"):
        assert c4[scheme]["bit_exact_vs_big_integers"] is True and c4[scheme]["commit_ms"] > 0 and c4[scheme]["proof_ms"] > 0
    assert c4["lagrange_shards_streamed"]["bit_exact_vs_big_integers"] is True and c4["lagrange_shards_streamed"]["commit_plus_proof_ms"] > 0
    sr = d["shard_rehearsal"]
    assert "error" not in sr and sr["bit_exact_vs_oracle"] is True and sr["exchange"].startswith("nccl")
    c4s = sr["config4_stream"]["sizes"]                  # the per-rank config-4 stream at the slice sizes of 1 / 2 / 4 / 8 ranks
    assert sorted(c4s) == ["2^11", "2^12", "2^13", "2^14"] and all(v["ms_per_blob"] > 0 and v["blobs"] == 24 for v in c4s.values())
    assert sorted(sr["sizes"]) == ["2^11", "2^12", "2^13"]
    for ranks, key in ((2, "2^13"), (4, "2^12"), (8, "2^11")):
        e = sr["sizes"][key]
        assert e["ranks"] == ranks and e["ms_per_step_20"] > 0 and e["ms_per_step_96"] > 0 and 0 < e["efficiency_96"] and e["bit_exact_vs_oracle"] is True
    sec = d["secondary"]
    assert sec["host_buffers_compute_proof_streamed_ms"] > 0 and sec["host_buffers_compute_proof_streamed_ifft_path_ms"] > 0
    assert sec["cached_lagrange_basis"]["host_buffers_compute_proof_ms"] > 0
    shapes = sec["reference_bench_shapes"]               # every criterion harness of the reference has its line: commit / proof / g1_ifft / verify
    for key in ("commit_coeff_512_ms", "compute_proof_512_ms", "g1_ifft_512_ms", "verify_proof_ms"):
        assert shapes[key] > 0, key
    # round 6: the rest of the reference's criterion suite (bench_kzg_setup, bench_kzg_commit_large_blobs; scaled with the reduced size here) with the
    # oracle port beside it, the blob -> commitment + proof stream, batch verification over >= 30 calls
    assert shapes["commit_8mb_ms"] > 0 and shapes["commit_16mb_ms"] > 0 and shapes["kzg_setup_%d_ms" % (1 << 13)] > 0
    cpu_shapes = sec["reference_bench_shapes_cpu_port"]
    assert cpu_shapes["commit_8mb_ms"] > 0 and cpu_shapes["commit_16mb_ms"] > 0 and cpu_shapes["kzg_setup_%d_ms" % (1 << 13)] > 0
    assert sec["commit_and_prove_blob_streamed_ms"] > 0 and sorted(sec["commit_and_prove_blob_streamed_by_jobs_in_flight_ms"]) == ["12", "8"]
    bv = sec["batch_verify_4096_end_to_end_stats"]
    assert bv["calls"] >= 30
    # where the kernel keeps them: the throttled periods of the process's cgroup and the slowest call's longest runqueue wait ride with the statistics
    if "calls_in_a_throttled_period" in bv:
        assert 0 <= bv["calls_in_a_throttled_period"] <= bv["calls"] and isinstance(bv["slowest_call_throttled"], bool)
    if "slowest_call_longest_runqueue_wait_ms" in bv:
        assert bv["slowest_call_longest_runqueue_wait_ms"] >= 0 and bv["median_call_longest_runqueue_wait_ms"] >= 0
    clk = d["gpu_clock_under_load"]                      # sysfs engine clock sampled in the untimed spin-up (None where sysfs does not show this GPU)
    assert clk is None or (200 < clk["sclk_mhz_min"] <= clk["sclk_mhz_mean"] <= clk["sclk_mhz_max"] < 4000 and clk["samples"] >= 1)


def test_bench_n_gpus_without_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher (no WORLD_SIZE) must produce the line: bench.py starts the two
    ranks itself before anything touches the GPU (one child process per rank, RANK / WORLD_SIZE / MASTER_* set), relays rank 0's JSON
    line and exits with the children's worst code.  On this one-GPU box the exchange is rehearsed over gloo, both ranks on GPU 0."""
    env = dict(os.environ, KZG_BENCH_LOG_N="16", KZG_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(key, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-secondary"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 6 and d["config"]["bit_exact_vs_oracle"] is True
    assert len(d["ms_per_step_per_rank"]) == 2 and all(v > 0 for v in d["ms_per_step_per_rank"])
    assert abs(max(d["ms_per_step_per_rank"]) - d["ms_per_step"]) < 1e-9            # MAX over ranks
    assert d["launched_by"].startswith("bench.py itself")


def test_bench_without_launcher_refuses_more_ranks_than_gpus_over_rccl():
    """One rank per GPU over RCCL needs N devices: on a box with fewer, `--gpus N` fails loudly BEFORE starting ranks (it must not
    measure N ranks on one GPU and call it N GPUs) unless the gloo rehearsal switch is set."""
    import torch
    n = torch.cuda.device_count() + 1
    env = dict(os.environ, KZG_BENCH_LOG_N="12")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KZG_BENCH_BACKEND", "MASTER_PORT"):
        env.pop(key, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode != 0 and "GPU(s) visible" in (res.stderr + res.stdout)


def test_bench_self_launch_propagates_a_failing_rank():
    """A rank that exits non-zero makes the launcher-less bench exit non-zero, and the surviving rank (waiting in the rendezvous for
    its peer) is ended after the grace period instead of hanging the run."""
    env = dict(os.environ, KZG_BENCH_LOG_N="12", KZG_BENCH_BACKEND="gloo", KZG_BENCH_RANK_GRACE_S="2", KZG_BENCH_FAIL_RANK="1")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(key, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-secondary"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 7, (res.returncode, res.stderr[-1500:])
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_bench_two_ranks_gloo_is_bit_exact():
    """The N > 1 path of bench.py rehearsed on one GPU: two ranks (gloo exchange, both on GPU 0), 2^18 pairs sharded by scalar
    index, the folded commitment checked on rank 0 against sum_i c_i tau^i * G1 (big-integer arithmetic)."""
    env = dict(os.environ, KZG_BENCH_LOG_N="18", KZG_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(key, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--no-secondary"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["bit_exact_vs_oracle"] is True
    assert d["config"]["steps_per_launch"] == 4          # 2^17 pairs per rank: four steps of the stream per batched launch (10 = 4 + 4 + 2)
    assert d["value"] > 0 and d["value_uniform"] > 0 and d["latency_ms"] > 0
    rep = d["replicas_mode"]                             # N > 1: whole commitments per rank beside the sharded headline (never `value`)
    assert rep["bit_exact_vs_oracle"] is True and rep["commitments_per_s"] > 0 and rep["error_on_rank_0"] is None


def test_bench_two_ranks_gloo_carries_config4():
    """at N > 1 the line times BASELINE config 4 (evaluations in host memory -> commitment and proof) through the
    evaluation-index shards of the Lagrange basis AND through round 4's replicated path, every result checked against big integers."""
    env = dict(os.environ, KZG_BENCH_LOG_N="15", KZG_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", KZG_BENCH_REPLICAS="0")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(key, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["bit_exact_vs_oracle"] is True
    c4 = d["config4"]
    assert c4["ok_on_every_rank"] is True, c4
    for scheme in ("lagrange_shards", "replicated_r4"):
        assert c4[scheme]["bit_exact_vs_big_integers"] is True and c4[scheme]["commit_ms"] > 0 and c4[scheme]["proof_ms"] > 0
    assert c4["lagrange_shards_streamed"]["bit_exact_vs_big_integers"] is True          # two blobs in flight per rank, exchanges over gloo
    assert "16384 / 2" not in c4["lagrange_shards"]["per_rank"] and "/ 2 elements" in c4["lagrange_shards"]["per_rank"]
    assert d["shard_rehearsal"] is None                   # the rehearsal belongs to the N = 1 line


def test_bench_one_rank_through_rccl_takes_the_multi_gpu_path():
    """What a one-GPU box can run of the driver's N > 1 bench over REAL RCCL (two ranks on one device are refused by RCCL): one rank with
    KZG_BENCH_FORCE_EXCHANGE=1 initialises the nccl process group and sends every step through the N > 1 code -- grouped launches
    (2^17 pairs: four steps per launch), bucketed all_gather_into_tensor of the XYZZ partials on the gatherer's own stream, host fold,
    the all-reduce / all-gather / broadcast around the timed region -- and every timed step is checked against sum_i c_i tau^i G1."""
    env = dict(os.environ, KZG_BENCH_LOG_N="17", KZG_BENCH_FORCE_EXCHANGE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KZG_BENCH_BACKEND", "MASTER_PORT"):
        env.pop(key, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3", "--no-secondary",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert [ln for ln in res.stdout.splitlines() if ln.strip()] == lines, res.stdout[:600]        # nothing but the line on stdout: RCCL's version banner goes to stderr
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["exchange_backend"] == "nccl"
    assert d["config"]["bit_exact_vs_oracle"] is True and d["config"]["steps_per_launch"] == 4
    assert d["value"] > 0 and d["value_uniform"] > 0 and d["latency_ms"] > 0


def test_bench_multi_mode_two_contexts_on_one_gpu_is_bit_exact():
    """`bench.py --multi --gpus 2` (one process, two kzg_multi contexts -- on this one-GPU box both on device 0): the line is well
    formed and every timed step's commitment matched its expected point (non-zero exit otherwise)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KZG_BENCH_LOG_N="17")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--multi", "--gpus", "2", "--steps", "9", "--warmup", "2"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-400:] + r.stderr[-800:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 9 and line["config"]["bit_exact_vs_oracle"] is True
    assert "roofline" in line and line["value"] > 0
