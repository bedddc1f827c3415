#!/bin/bash
# accumulate kernel at 2 / 3 / 4 waves per SIMD (variants built with -DKZG_ACC_WAVES, grid sized with KZG_ACC_SLOTS)
cd "$(dirname "$0")/.."
run() { # lib slots
  if [ -n "$1" ]; then export KZG_LIB_PATH=$PWD/$1; else unset KZG_LIB_PATH; fi
  KZG_ACC_SLOTS=$2 python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s slots %5s step %.4f lat %.4f acc alone %.4f piped %.4f exact %s' % ('${1:-in-tree(3)}', '$2', d['ms_per_step'], d['latency_ms'], r['avg_launch_ms'], r['avg_launch_ms_pipelined'], d['config']['bit_exact_vs_oracle']))"
}
run "" 3072; run gpurun_variants/libkzg_acc2.so 2048; run gpurun_variants/libkzg_acc4.so 4096; run "" 3072; run gpurun_variants/libkzg_acc2.so 2048; run gpurun_variants/libkzg_acc4.so 4096
