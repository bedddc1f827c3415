R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-icache}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export KZG_BENCH_PIPELINE=0
$R/tools/ubench/field_rates > $O/field_rates.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $O/a -o a -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $O/a.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $O/u -o u -- $R/tools/ubench/field_rates > $O/u.log 2>&1 || exit 4
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $O/u2 -o u2 -- $R/tools/ubench/field_rates > $O/u2.log 2>&1 || exit 5
python3 - <<PY
import csv, collections, glob
for d in ("a", "u", "u2"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("kzg::", "").replace("void ", "")[:26]
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            if "accumulate" in k or "madd" in k:
                print("%-28s %-24s n=%d avg=%.4g min=%.4g" % (k, c, len(v), sum(v) / len(v), min(v)))
PY
cat $O/field_rates.txt
