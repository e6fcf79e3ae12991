for h in 80 100 120; do echo "== hold $h"; CV2_FIRST_ROUND_HOLD_MS=$h python tools/bench_streams.py 8 10 --stagger 40 2>&1 | tail -1; done
CV2_FIRST_ROUND_HOLD_MS=100 python tools/bench_streams.py 8 3 --stagger 40 --trace > gpurun_out/r5_stagger_trace_hold100.txt 2>&1
