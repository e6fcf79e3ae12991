#!/bin/bash
# Round 6: the 17..32-row decode step's attention as ONE block per (row, kv head) that walks all key tiles and leaves the O projection's operand
# planes itself (CV2_ATT_ONE_SPLIT=1) against key splits + a combine launch (default).  Step time at 32 / 24-row launches and greedy ids.
cd "$(dirname "$0")/.."
for v in 0 1 0 1; do
  echo "== CV2_ATT_ONE_SPLIT=$v"
  CV2_ATT_ONE_SPLIT=$v python tools/bench_rows.py 32 28 2>&1 | grep -v amdgpu.ids
done
CV2_ATT_ONE_SPLIT=1 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "b32_fr_de_ids or ragged_live_rows" 2>&1 | tail -2
