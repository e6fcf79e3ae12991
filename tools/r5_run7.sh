python -m pytest tests/test_hift_gpu.py tests/test_api_gpu.py -q -m gpu -x 2>&1 | tail -8
for h in 0 70; do echo "== hold $h"; CV2_FIRST_ROUND_HOLD_MS=$h python tools/bench_streams.py 8 10 --stagger 40 2>&1 | tail -2; CV2_FIRST_ROUND_HOLD_MS=$h python tools/bench_streams.py 8 4 2>&1 | tail -2; done
for v in 1 0; do echo "sd_scalar=$v"; CV2_HIFT_SD_SCALAR=$v python tools/bench_hift.py 500 20 2>&1 | tail -1; done
