# Per-kernel durations of any python tool: tools/prof_stats_cmd.sh <tag> <script.py> [args...] -> gpurun_out/<tag>_kernel_stats.csv (+ top 24 printed)
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o run -- python3 $SCRIPT "$@" > $ROOT/gpurun_out/${TAG}_run.log 2>&1
tail -2 $ROOT/gpurun_out/${TAG}_run.log | grep -v simple_timer
f=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
cp "$f" $ROOT/gpurun_out/${TAG}_kernel_stats.csv
cp "$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)" $ROOT/gpurun_out/${TAG}_kernel_trace.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-30s calls %4s  avg %9.1f us  min %9.1f us  max %9.1f us" % (r["Name"].split("(")[0][-30:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
