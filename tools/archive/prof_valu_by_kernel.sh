# VALU instructions and active cycles per kernel of one 2^20 commitment (rocprofv3 --pmc, one commitment at a time): where the
# integer issue slots of a pipelined step go besides k_msm_accumulate.   usage (GPU box): bash tools/prof_valu_by_kernel.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-valu}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export KZG_BENCH_PIPELINE=0
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/sq -o sq -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $O/sq.log 2>&1 || exit 3
python3 - <<PY
import csv, collections, glob
for f in glob.glob("$O/sq/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("kzg::", "").replace("void ", "")
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    kernels = sorted({k for k, _ in agg})
    print("%-34s %8s %14s %14s %12s %12s" % ("kernel (average per dispatch)", "calls", "SQ_INSTS_VALU", "ACTIVE_VALU", "INSTS_LDS", "WAVES"))
    rows = []
    for k in kernels:
        g = lambda c: (sum(agg[(k, c)]) / len(agg[(k, c)])) if agg.get((k, c)) else 0.0
        rows.append((g("SQ_INSTS_VALU"), k, len(agg[(k, "SQ_INSTS_VALU")]), g("SQ_ACTIVE_INST_VALU"), g("SQ_INSTS_LDS"), g("SQ_WAVES")))
    for v, k, n, a, l, w in sorted(rows, reverse=True)[:16]:
        print("%-34s %8d %14.4g %14.4g %12.4g %12.4g" % (k[:34], n, v, a, l, w))
PY
