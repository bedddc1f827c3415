R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-acclevel}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export KZG_BENCH_PIPELINE=0
timeout -k 10 300 rocprofv3 --pmc SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $O/a -o a -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $O/a.log 2>&1
python3 - <<PY
import csv, collections, glob
for f in glob.glob("$O/a/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].replace("kzg::", "")[:26], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        if "accumulate" in k: print("%-28s %-24s n=%d avg=%.4g" % (k, c, len(v), sum(v) / len(v)))
PY
