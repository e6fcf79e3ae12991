#!/bin/bash
# A/B on one box: B=32 bench of the tree at an older commit (extracted + built under _ab_old/) against the current tree
# setup (here, before the gpurun call): rm -rf _ab_old; mkdir _ab_old; git archive <commit> | tar -x -C _ab_old; (cd _ab_old && ./build.sh)
cd "$(dirname "$0")/.."
for rep in 1 2; do
for d in _ab_old .; do
  (cd $d && python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', d['value'], d['stages']['ms_per_step'])")
done
done
