# kernel trace of tools/step_series2.py (LENS=2,5): where is the one-time stall of the second stream?
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_st
export LENS=2,5,5
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_st -o run -- python3 $ROOT/tools/step_series2.py > $ROOT/gpurun_out/st_run.log 2>&1
grep stream $ROOT/gpurun_out/st_run.log
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_st/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("kzg::", "")[:28], r.get("Queue_Id", ""), r.get("Stream_Id", "")) for r in rows]
# find first accumulate; print everything from the 2nd accumulate's end to the 5th accumulate
acc = [i for i, k in enumerate(ks) if "accumulate" in k[2]]
t0 = ks[acc[0]][0]
for i in range(acc[0] - 8, min(len(ks), acc[4] + 4)):
    s, e, n, q, st = ks[i]
    print("%9.3f ms  +%8.3f ms  %-28s q=%s s=%s  gap before %.3f ms" % ((s - t0) / 1e6, (e - s) / 1e6, n, q, st, (s - ks[i - 1][1]) / 1e6 if i else 0))
PY
