# A/B of the staggered start of an MSM launch behind the previous launch's sort (KZG_MSM_STAGGER=1 default / 0): per-step time of K-step
# regions at shard sizes and at 2^20, short (the driver's 20 steps) and long
for cfg in "17 20" "18 20" "19 20" "20 20" "17 96" "20 96"; do set -- $cfg; for v in 1 0; do echo -n "2^$1 pairs, $2 steps, stagger=$v: "; for i in 1 2 3 4; do KZG_MSM_STAGGER=$v KZG_BENCH_LOG_N=$1 python bench.py --gpus 1 --steps $2 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['ms_per_step'],4), end=' ' if d['config']['bit_exact_vs_oracle'] else ' WRONG ')"; done; echo; done; done
