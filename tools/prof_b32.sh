cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_b32d
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_b32d -- python3 $GRAFT_REPO_ROOT/bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_b32d.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py gpurun_out/prof_b32d 2>&1 | head -40
find gpurun_out/prof_b32d -name '*_kernel_trace.csv' -delete     # tens of MB; the stats CSV and the summary are what is kept
tail -1 gpurun_out/prof_b32d.log | cut -c1-400
