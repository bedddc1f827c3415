# rocprofv3 kernel stats of bench.py with ONE commitment in flight (KZG_BENCH_PIPELINE=0): every k_msm_accumulate launch runs alone, so the
# average duration is directly comparable with roofline.avg_launch_ms of the bench line -> gpurun_out/<tag>_depth1_kernel_stats.csv
TAG=${1:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_d1
export KZG_BENCH_PIPELINE=0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d1 -o run -- python3 $ROOT/bench.py --steps 30 --warmup 4 --no-secondary --no-cpu-baseline > $ROOT/gpurun_out/${TAG}_depth1_bench.log 2>&1
cp "$(find /tmp/prof_d1 -name '*kernel_stats.csv' | head -1)" $ROOT/gpurun_out/${TAG}_depth1_kernel_stats.csv
grep -o '"avg_launch_ms": [0-9.]*' $ROOT/gpurun_out/${TAG}_depth1_bench.log
head -4 $ROOT/gpurun_out/${TAG}_depth1_kernel_stats.csv | cut -c1-60,200-
