# A/B of the knobs a LONE mid-size MSM has: wave slots of the accumulate grid, NAF width, event polling.
# usage (GPU box): bash tools/ab_lone_msm.sh > gpurun_out/r05_ab_lone.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
run() { echo "== $*"; env "$@" python $ROOT/tools/time_lone_msm.py 18 19 2>/dev/null | grep -v amdgpu.ids; }
run KZG_NOOP=1
run KZG_EVENT_POLL=1
run KZG_ACC_SLOTS=2048
run KZG_ACC_SLOTS=4096
run KZG_NAF_C=15
run KZG_NAF_C=17
run KZG_NAF_C=15 KZG_ACC_SLOTS=2048 KZG_EVENT_POLL=1
run KZG_QUAD_REDUCE=0
