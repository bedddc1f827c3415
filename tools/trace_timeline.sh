# kernel timeline of the pipelined bench (two MSMs in flight): who runs beside whom
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-timeline}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary > $O/kt.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("kzg::", "").split("(")[0], r.get("Queue_Id", r.get("Stream_Id", "?"))))
rows.sort()
# last ~2.2 steps
acc = [r for r in rows if r[2] == "k_msm_accumulate"]
t0 = acc[-4][0]
for s, e, n, q in rows:
    if s >= t0 - 200000 and s <= acc[-1][1]:
        print("%9.1f %9.1f %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
PY
