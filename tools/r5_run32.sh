python tools/bench_streams.py 8 10 --stagger 40 2>&1 | tail -1
python tools/bench_streams.py 4 10 --stagger 40 2>&1 | tail -1
python tools/bench_streams.py 4 3 --stagger 40 --trace > gpurun_out/r5_stagger4_trace.txt 2>&1
