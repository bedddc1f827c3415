# sweep: steps per launch x bucket bits of the batched table mode at shard sizes (KZG_SHARD_GROUP, KZG_BATCH_C)
for cfg in "17 4 15" "17 8 14" "17 8 13" "18 2 16" "18 4 15" "18 8 14" "19 2 16" "19 4 15"; do set -- $cfg; echo -n "log_n=$1 group=$2 c=$3: "; KZG_BENCH_LOG_N=$1 KZG_SHARD_GROUP=$2 KZG_BATCH_C=$3 python bench.py --no-secondary --no-cpu-baseline --steps 96 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['config'].get('steps_per_launch'), d['config'].get('pipeline_depth'), d['config']['bit_exact_vs_oracle'])"; echo; done
