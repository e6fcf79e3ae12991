#!/bin/bash
# streaming extra of bench.py with and without the per-call flow cache
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/stream_cache_on.json 2> gpurun_out/stream_cache_on.err
CV2_FLOW_CACHE=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/stream_cache_off.json 2> gpurun_out/stream_cache_off.err
python - <<'P'
import json
for f in ('on','off'):
    try:
        d=json.loads(open(f'gpurun_out/stream_cache_{f}.json').read().strip().splitlines()[-1])
        print(f, json.dumps(d['extra']['streaming'], indent=None)[:1500])
    except Exception as e:
        print(f, 'ERR', e); print(open(f'gpurun_out/stream_cache_{f}.err').read()[-3000:])
P
