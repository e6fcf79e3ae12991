# Round 6: bench.py (B=1 + extras, no CPU baseline) with the scheduler's newcomer_beside + first_chunk_lane switches off (0) and on (1, the default), twice.
for rep in 1 2; do
for v in 0 1; do
CV2_NEWCOMER_BESIDE=$v CV2_FIRST_CHUNK_LANE=$v CV2_BENCH_LIVE_PMC=0 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['extra']['streaming']
print('beside+lane=$v: B=1 %.1f; batch32 %.1f; streams_1 %s' % (d['value'], d['extra']['batch32']['value'], s['streams_1']))
print('   streams_8 %s' % s['streams_8'])
print('   bistream_8 %s' % {k: v for k, v in s['bistream_8'].items() if k != 'workload'})"
done; done
