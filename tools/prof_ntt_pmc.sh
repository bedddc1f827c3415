R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-nttpmc}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > $O/run.py <<PY
import sys, ctypes as C
sys.path.insert(0, "$R")
import numpy as np, torch
import rust_kzg_bn254_amd as k
from rust_kzg_bn254_amd import _lib
lib = _lib.load(); ctx = k.Context(0)
n = 1 << 20
a = np.random.default_rng(1).integers(0, 1 << 60, size=(n, 4), dtype=np.uint64)
d = torch.from_numpy(a.view(np.int64)).cuda(); torch.cuda.synchronize()
for i in range(20):
    lib.kzg_fr_ntt_device(ctx.handle, C.c_void_p(d.data_ptr()), n, i & 1)
torch.cuda.synchronize()
PY
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $O/run.py > $O/kt.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/sq -o sq -- python3 $O/run.py > $O/sq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/sq2 -o sq2 -- python3 $O/run.py > $O/sq2.log 2>&1
python3 - <<PY
import csv, collections, glob
for d in ("sq", "sq2"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("kzg::", "")[:12]
            if "ntt_pass" in r["Kernel_Name"]:
                agg[(r["Counter_Name"])].append(float(r["Counter_Value"]))
        for c, v in sorted(agg.items()):
            # three passes per transform: report per pass position
            per = [sum(v[i::2]) / len(v[i::2]) for i in range(2)] + [0]
            print("%-24s pass1=%.4g pass2=%.4g pass3=%.4g" % (c, per[0], per[1], per[2]))
for f in glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True):
    for ln in open(f):
        if "ntt" in ln or "Name" in ln: print(ln.strip()[:160])
PY
