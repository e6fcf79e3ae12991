# rocprofv3 kernel trace of the default bench (batch 1); summary printed by tools/prof_summary.py
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_b1d
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_b1d -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/gpurun_out/prof_b1d.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py gpurun_out/prof_b1d 2>&1 | head -45
find gpurun_out/prof_b1d -name '*_kernel_trace.csv' -delete     # tens of MB; the stats CSV and the summary are what is kept
tail -1 gpurun_out/prof_b1d.log | cut -c1-300
