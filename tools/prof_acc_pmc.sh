# SQ counter pass over the accumulate kernel: where do the wave cycles go (issue / wait / active) and what clock does the chip hold.
# usage (GPU box): bash tools/prof_acc_pmc.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-accpmc}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export KZG_BENCH_PIPELINE=0
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $O/kt.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/sq -o sq -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $O/sq.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $O/sq2 -o sq2 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $O/sq2.log 2>&1 || exit 4
python3 - <<PY
import csv, collections, glob
for d in ("sq", "sq2"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("kzg::", "")
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            if "accumulate" in k or "bucket" in k:
                print("%-28s %-24s n=%d avg=%.4g" % (k, c, len(v), sum(v) / len(v)))
for f in glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True):
    print(open(f).read())
PY
