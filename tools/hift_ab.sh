#!/bin/bash
# HiFT convolution kernels A/B: bf16 x 6 split products (k_conv6, default) against the fp32 matrix-core kernel everywhere (CV2_HIFT_FP32=1)
cd "$(dirname "$0")/.."
for v in 0 1; do
  echo "CV2_HIFT_FP32=$v"
  CV2_HIFT_FP32=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(' B=1', d['value'], d['stages']['ms_per_step'])
print(' B=32', d['extra']['batch32']['value'], d['extra']['batch32']['stages_ms'])
s=d['extra']['streaming']; print(' streams_1', s['streams_1']); print(' streams_8', s['streams_8'])"
done
