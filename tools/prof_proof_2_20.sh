# rocprofv3 --kernel-trace --stats of each proof / blob entry point at 2^20, one run per entry point:
#   gpurun_out/<tag>_proof/<what>_kernel_stats.csv  +  <what>.log (wall time per call)
# usage (GPU box): bash tools/prof_proof_2_20.sh r05 "proof_off proof_on commit_eval commit_blob proof_stream commit_stream"
TAG=${1:-r05}
WHATS=${2:-"proof_off proof_on commit_eval commit_blob proof_stream commit_stream"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/${TAG}_proof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in $WHATS; do
  rm -rf /tmp/prof_pp
  export WHAT=$W
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pp -o run -- python3 $ROOT/tools/prof_proof_2_20.py > $O/$W.log 2>&1 || { echo "$W failed"; tail -5 $O/$W.log; exit 1; }
  cp "$(find /tmp/prof_pp -name '*kernel_stats.csv' | head -1)" $O/${W}_kernel_stats.csv
  grep "ms per call" $O/$W.log
done
# the per-rank calls of config 4 over a 2^17-element Lagrange shard (csrc/lagrange.hip): gpurun_out/<tag>_proof/rank8_kernel_stats.csv
if [ -z "$SKIP_RANK8" ]; then
  rm -rf /tmp/prof_pp
  WHAT=rank8 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pp -o run -- python3 $ROOT/tools/time_config4_shards.py > $O/rank8.log 2>&1 || { echo "rank8 failed"; tail -5 $O/rank8.log; exit 1; }
  cp "$(find /tmp/prof_pp -name '*kernel_stats.csv' | head -1)" $O/rank8_kernel_stats.csv
  grep "lagrange shard" $O/rank8.log
fi
