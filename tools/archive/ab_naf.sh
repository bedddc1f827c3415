#!/bin/bash
# same-box A/B of the NAF mode (per-bit tables) against the fixed-window tables: alternating runs of bench.py
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for v in "KZG_NAF_OFF=0" "KZG_NAF_OFF=1" ${AB_EXTRA}; do
    env $v python bench.py --steps ${AB_STEPS:-40} --warmup 6 --no-secondary --no-cpu-baseline 2>gpurun_out/ab_err.log > gpurun_out/ab_out.log || { tail -5 gpurun_out/ab_err.log; exit 1; }
    python - "$v" <<'PY'
import sys, json
d = json.loads(open("gpurun_out/ab_out.log").read().strip().splitlines()[-1])
r = d["roofline"]; p = d["phases_ms_per_launch"]
print("%-16s step %.4f  latency %.4f  acc alone %.4f  pipelined %.4f  sort %.3f  red1 %.3f red2 %.3f exact %s" % (sys.argv[1], d["ms_per_step"], d["latency_ms"], r["avg_launch_ms"], r["avg_launch_ms_pipelined"], p["digits_or_coarse_hist"] + p["sort_pass1"] + p["sort_pass2"], p["bucket_sums_reduce1"], p["reduce2"], d["config"]["bit_exact_vs_oracle"]), flush=True)
PY
  done
done
