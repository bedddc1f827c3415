#!/bin/bash
# same-box A/B of library variants through the whole bench line: tools/ab_full.sh <variant.so|""> ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  if [ -n "$v" ]; then export KZG_LIB_PATH=$PWD/$v; else unset KZG_LIB_PATH; fi
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phases_ms_per_launch']; s=d['secondary']
print('%-32s step %.4f lat %.4f acc %.4f b1 %.4f b2 %.4f | ntt %.4f intt %.4f proof %.3f commit_blob %.3f bv_core %.3f bv_e2e %.3f exact %s' % ('${v:-in-tree}', d['ms_per_step'], d['latency_ms'], p['accumulate'], p['bucket_sums_reduce1'], p['reduce2'], s['fr_ntt_ms'], s['fr_intt_ms'], s['host_buffers_compute_proof_ms'], s['commit_blob_from_host_bytes_ms'], s['batch_verify_4096_core_ms'], s['batch_verify_4096_end_to_end_ms'], d['config']['bit_exact_vs_oracle']))"
done
