# end-of-round measurement batch: kernel traces of the B=1 and B=32 bench, the default bench line (with its extras), PMC traffic of the decode step
bash tools/prof_b1.sh > gpurun_out/final_prof_b1.txt 2>&1
bash tools/prof_b32.sh > gpurun_out/final_prof_b32.txt 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/final_bench_b1.json 2> gpurun_out/final_bench_b1.err
python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/final_bench_b32.json 2> gpurun_out/final_bench_b32.err
tail -1 gpurun_out/final_bench_b1.json | cut -c1-250; tail -1 gpurun_out/final_bench_b32.json | cut -c1-250
