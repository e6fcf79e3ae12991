# end-of-round measurement batch (one box, one run): PMC counters per stage, kernel traces of the B=1 and B=32 bench, the chain timeline,
# the row sweep, then the default bench line (with its extras; it reads the PMC files just collected) and the B=32 line
R=$GRAFT_REPO_ROOT
for s in decode hift flow; do bash tools/pmc_stages.sh $s $([ $s = decode ] && echo 1 || echo 3) > gpurun_out/final_pmc_$s.txt 2>&1; done
cp gpurun_out/r6_pmc_decode.json gpurun_out/r6_pmc_hift.json gpurun_out/r6_pmc_flow.json profiles/ 2>/dev/null
bash tools/prof_b1.sh > gpurun_out/final_prof_b1.txt 2>&1
bash tools/prof_b32.sh > gpurun_out/final_prof_b32.txt 2>&1
cd $R
CV2_AMD_LIB=$R/cosyvoice2-eu_amd/cv2amd/libcv2amd_dbg.so python tools/dbg_chain.py > gpurun_out/final_chain_timeline.txt 2>&1
python tools/bench_rows_sweep.py > gpurun_out/final_rows_sweep.txt 2>&1
bash tools/prof_stream.sh 8 > gpurun_out/final_stream8.txt 2>&1
python tools/bench_threads.py 8 3 > gpurun_out/final_threads8.txt 2>&1
python bench.py > gpurun_out/final_bench_b1.json 2> gpurun_out/final_bench_b1.err
python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/final_bench_b32.json 2> gpurun_out/final_bench_b32.err
tail -1 gpurun_out/final_bench_b1.json | cut -c1-250; tail -1 gpurun_out/final_bench_b32.json | cut -c1-250
