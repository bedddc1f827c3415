# Same-box step times of bench.py at the shard sizes of a 2^20-pair commitment over 1 / 2 / 4 / 8 ranks (one rank here: what each rank of
# an N-rank run computes, without the exchange) -> the strong-scaling prediction quoted in DESIGN.md section 7.
for ln in 20 19 18 17; do echo -n "2^$ln pairs per rank: "; KZG_BENCH_LOG_N=$ln python bench.py --no-secondary --no-cpu-baseline --steps 96 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%.4f ms per step, %d steps per launch, depth %d, bit-exact %s' % (d['ms_per_step'], d['config'].get('steps_per_launch'), d['config'].get('pipeline_depth'), d['config']['bit_exact_vs_oracle']))"; done
