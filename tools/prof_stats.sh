# Per-kernel durations of a short bench run: tools/prof_stats.sh [tag] -> gpurun_out/<tag>_kernel_stats.csv (printed, top 20)
TAG=${1:-stats}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o run -- python3 $ROOT/bench.py --steps 20 --warmup 4 --no-secondary --no-cpu-baseline > $ROOT/gpurun_out/${TAG}_bench.log 2>&1
f=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
cp "$f" $ROOT/gpurun_out/${TAG}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:20]:
    print("%-28s calls %4s  avg %9.1f us  min %9.1f us  max %9.1f us" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
