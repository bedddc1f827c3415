# rocprofv3 kernel stats of tools/archive/prof_proof.py (10 proofs + 10 eval-form commitments) at LOG_N = $1..: top kernels by time
cd /tmp && export TMPDIR=/tmp
for ln in "$@"; do
  rm -rf /tmp/pp$ln
  LOG_N=$ln rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp$ln -o run -- python3 $GRAFT_REPO_ROOT/tools/archive/prof_proof.py > /tmp/pp$ln.log 2>&1
  echo "== 2^$ln"; grep "ms$" /tmp/pp$ln.log || tail -3 /tmp/pp$ln.log
  LN=$ln python3 - <<'PY'
import csv, glob, os
fs = glob.glob("/tmp/pp%s/**/*kernel_stats.csv" % os.environ["LN"], recursive=True)
if fs:
    for r in list(csv.DictReader(open(fs[0])))[:16]:
        print("   %-44s calls %4s avg %8.1f us  %5s %%" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
