R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-variants}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/tools/ubench/acc_variants 80 > $O/variants.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/v -o v -- $R/tools/ubench/acc_variants 80 > $O/v.log 2>&1
python3 - <<PY
import csv, collections, glob
dur = collections.defaultdict(list)
for f in glob.glob("$O/v/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:40]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for f in glob.glob("$O/v/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    ks = sorted(set(k for k, _ in agg))
    for k in ks:
        g = {c: sum(v) / len(v) for (kk, c), v in agg.items() if kk == k}
        d = sum(dur[k]) / max(1, len(dur[k]))
        clk = g.get("GRBM_GUI_ACTIVE", 0) / 8 / max(d, 1)
        wc = g.get("SQ_WAVE_CYCLES", 1)
        print("%-40s dur=%.3f ms clock=%.2f GHz insts/wave=%.0f cyc/inst/SIMD=%.2f wait_any=%.1f%% wait_inst=%.1f%% active=%.1f%%" % (
            k, d / 1e6, clk, g.get("SQ_INSTS_VALU", 0) / 3072, g.get("GRBM_GUI_ACTIVE", 0) / 8 / (g.get("SQ_INSTS_VALU", 1) / 1024),
            100 * g.get("SQ_WAIT_ANY", 0) / wc, 100 * g.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * g.get("SQ_ACTIVE_INST_ANY", 0) / wc))
PY
cat $O/variants.txt
