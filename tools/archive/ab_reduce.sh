#!/bin/bash
# same-box A/B of the two bucket-reduction levels: one-lane kernels (KZG_PAIR_REDUCE=0) against the lane-pair kernels
# (curve_pair.h): phase times of the 2^20 MSM from bench.py, small commitments and shard-sized MSMs.
set -o pipefail
out=gpurun_out/r03_ab_reduce.txt
: > $out
for mode in 0 1 0 1; do
  echo "== KZG_PAIR_REDUCE=$mode" >> $out
  KZG_PAIR_REDUCE=$mode python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
p=d['phases_ms_per_launch']
print('ms_per_step %.4f latency %.4f acc %.4f bits1 %.4f bits2 %.4f total %.4f exact %s' % (d['ms_per_step'], d['latency_ms'], p['accumulate'], p['bucket_sums_reduce1'], p['reduce2'], p['device_total'], d['config']['bit_exact_vs_oracle']))" >> $out
  KZG_PAIR_REDUCE=$mode python tools/time_commit_sizes.py 2>/dev/null | tail -12 >> $out
done
cat $out
