for cfg in "17 2" "17 4" "17 8" "18 1" "18 2" "18 4" "19 1" "19 2"; do set -- $cfg; echo -n "log_n=$1 group=$2: "; KZG_BENCH_LOG_N=$1 KZG_SHARD_GROUP=$2 python bench.py --no-secondary --no-cpu-baseline --steps 96 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['config'].get('steps_per_launch'), d['config'].get('pipeline_depth'), d['config']['bit_exact_vs_oracle'])"; done
