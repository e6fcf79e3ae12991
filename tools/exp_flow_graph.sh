#!/bin/bash
# Round 6: cv2_flow_inference as ONE hipGraph launch per repeated shape (CV2_FLOW_GRAPH=1; default off) against the ~2 300 launches per call.
# The headline line without extras / CPU baseline, both ways, twice.   bash tools/exp_flow_graph.sh
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for g in 0 1; do
    CV2_FLOW_GRAPH=$g CV2_BENCH_LIVE_PMC=0 python bench.py --no-extra --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('CV2_FLOW_GRAPH=$g: %.2f audio-s/s, %.2f ms per utterance; stages %s' % (d['value'], d['ms_per_step'], d['stages']['ms_per_step']))"
  done
done
