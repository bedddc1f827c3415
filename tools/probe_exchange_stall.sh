# Diagnostics for the one-off stall ~40 steps into a timed region when every step goes through the exchange (one rank rehearsing the
# N > 1 path: KZG_BENCH_FORCE_EXCHANGE=1).  Cause found in round 4: CPython's cyclic collector (profiles/r04_exchange_gc_stall.txt);
# bench.py now switches it off around its measurements (KZG_BENCH_GC=1 leaves it on).  Prints the value and the gaps > 1.6 ms.
run() { tag=$1; shift; env "$@" KZG_BENCH_TRACE=1 python bench.py --no-secondary --no-cpu-baseline --steps 120 2>gpurun_out/stall_$tag.err | grep value | cut -c95-130; grep "steps(120" gpurun_out/stall_$tag.err | tr ' ' '\n' | awk '$1+0 > 1.6 || NR < 4' | tr '\n' ' '; echo; }
run force_b1_gc KZG_BENCH_FORCE_EXCHANGE=1 KZG_BENCH_EXCHANGE_BUCKET=1 KZG_BENCH_GC=1
run force_b1 KZG_BENCH_FORCE_EXCHANGE=1 KZG_BENCH_EXCHANGE_BUCKET=1
run force_b8 KZG_BENCH_FORCE_EXCHANGE=1
run plain X=1
