# end-of-round measurement batch: kernel traces of the B=1 and B=32 bench, the bench lines themselves, streaming latency
bash tools/prof_b1.sh > gpurun_out/final_prof_b1.txt 2>&1
bash tools/prof_b32.sh > gpurun_out/final_prof_b32.txt 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/final_bench_b1.json 2> gpurun_out/final_bench_b1.err
python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/final_bench_b32.json 2> gpurun_out/final_bench_b32.err
python tools/bench_stream.py > gpurun_out/final_stream.txt 2>&1
tail -1 gpurun_out/final_bench_b1.json | cut -c1-250; tail -1 gpurun_out/final_bench_b32.json | cut -c1-250; tail -3 gpurun_out/final_stream.txt
