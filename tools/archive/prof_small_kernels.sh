cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  rm -rf /tmp/ps$n
  N=$n rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps$n -o run -- python3 $GRAFT_REPO_ROOT/tools/archive/prof_small_shapes.py > /tmp/ps$n.log 2>&1
  echo "== n = $n"; grep "ms$" /tmp/ps$n.log || tail -3 /tmp/ps$n.log
  N=$n python3 - <<'PY'
import csv, glob, os
fs = glob.glob("/tmp/ps%s/**/*kernel_stats.csv" % os.environ["N"], recursive=True)
if fs:
    for r in list(csv.DictReader(open(fs[0]))):
        if int(r["Calls"]) >= 60: print("   %-44s calls %4s avg %7.1f us" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
