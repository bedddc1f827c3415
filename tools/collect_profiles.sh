set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${PROF_TAG:-v4}; mkdir -p $O
test -n "$SKIP_BENCH" || (cd $R && timeout -k 10 280 python bench.py > $O/bench_line.json 2> $O/bench.err) || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/kt.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_write.log 2>&1 || exit 4
ls -la $O $O/kt $O/pmc_fetch $O/pmc_write
tail -c 600 $O/bench_line.json

# summaries for profiles/ (copy them there and commit): kernel stats of the bench run + PMC traffic per kernel
F=$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_write -name '*counter_collection.csv' | head -1)
python3 $R/tools/pmc_summarize.py $F $W $O/pmc_summary.json 20
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv
