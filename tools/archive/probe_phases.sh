#!/bin/bash
# phase times of one 2^20 MSM (bench.py, one at a time) for library variants: tools/probe_phases.sh <variant.so|""> ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  if [ -n "$v" ]; then export KZG_LIB_PATH=$PWD/$v; else unset KZG_LIB_PATH; fi
  python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phases_ms_per_launch']
print('%-34s acc %.4f bits1 %.4f bits2 %.4f total %.4f' % ('${v:-in-tree}', p['accumulate'], p['bucket_sums_reduce1'], p['reduce2'], p['device_total']))"
done
