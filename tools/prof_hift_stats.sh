# kernel stats of the vocoder alone (500 frames, 30 calls)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_hs
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_hs -- python3 $GRAFT_REPO_ROOT/tools/bench_hift.py 500 30 > $GRAFT_REPO_ROOT/gpurun_out/prof_hs.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py gpurun_out/prof_hs 2>&1 | head -24
find gpurun_out/prof_hs -name '*_kernel_trace.csv' -delete
