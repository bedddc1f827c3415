# Same-box A/B of library variants through bench.py: tools/ab_bench.sh <variant.so|""> ...   ("" = the in-tree build)
# prints ms_per_step, latency, accumulate launch alone / pipelined, bit-exact flag per run
cd "$(dirname "$0")/.."
for v in "$@"; do
  if [ -n "$v" ]; then export KZG_LIB_PATH=$PWD/$v; else unset KZG_LIB_PATH; fi
  python bench.py --steps ${AB_STEPS:-40} --warmup 6 --no-secondary --no-cpu-baseline 2>gpurun_out/ab_err.log > gpurun_out/ab_out.log || { tail -5 gpurun_out/ab_err.log; exit 1; }
  python - "$v" <<'PY'
import sys, json
d = json.loads(open("gpurun_out/ab_out.log").read().strip().splitlines()[-1])
r = d["roofline"]
print("lib=%-36s step %.4f  latency %.4f  acc alone %.4f  acc pipelined %.4f  exact %s" % (sys.argv[1] or "(in-tree)", d["ms_per_step"], d["latency_ms"], r["avg_launch_ms"], r["avg_launch_ms_pipelined"], d["config"]["bit_exact_vs_oracle"]))
PY
done
