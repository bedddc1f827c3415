# HBM traffic (rocprofv3 --pmc, one counter per pass) of tools/ubench/batch_affine: bytes per addition of every variant.
# Run on the GPU box from the repository root: bash tools/prof_batch_affine_pmc.sh [table_log]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O; T=${1:-24}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/ba_fetch_$T -o f -- $R/tools/ubench/batch_affine $T 48 > $O/ba_fetch_$T.log 2>&1 || exit 3
timeout -k 10 250 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/ba_write_$T -o w -- $R/tools/ubench/batch_affine $T 48 > $O/ba_write_$T.log 2>&1 || exit 4
F=$(find $O/ba_fetch_$T -name '*counter_collection.csv' | head -1); W=$(find $O/ba_write_$T -name '*counter_collection.csv' | head -1)
python3 $R/tools/pmc_summarize.py $F $W $O/ba_pmc_$T.json $T
cat $O/ba_pmc_$T.json
