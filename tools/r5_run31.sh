python tools/bench_streams.py 8 10 --stagger 40 2>&1 | tail -1
python tools/bench_streams.py 2 10 --stagger 40 2>&1 | tail -1
CV2_FIRST_ROUND_HOLD_MS=0 python tools/bench_streams.py 2 10 --stagger 40 2>&1 | tail -1
python tools/bench_streams.py 4 10 --stagger 40 2>&1 | tail -1
CV2_FIRST_ROUND_HOLD_MS=0 python tools/bench_streams.py 4 10 --stagger 40 2>&1 | tail -1
