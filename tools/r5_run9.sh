python tools/bench_streams.py 8 10 --stagger 40 2>&1 | tail -1
python tools/bench_streams.py 8 4 2>&1 | tail -2
python tools/bench_streams.py 8 3 --stagger 40 --trace > gpurun_out/r5_stagger_trace_pieces.txt 2>&1
python -m pytest tests/test_api_gpu.py -q -m gpu -x 2>&1 | tail -3
