# kernel trace of the streaming chunk path (tools/prof_stream.py) + how much flow / HiFT / decode overlap
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_stream
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_stream -- python3 $GRAFT_REPO_ROOT/tools/prof_stream.py ${1:-1} > $GRAFT_REPO_ROOT/gpurun_out/prof_stream.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py gpurun_out/prof_stream 2>&1 | head -${2:-40}
python tools/prof_overlap.py gpurun_out/prof_stream
find gpurun_out/prof_stream -name '*_kernel_trace.csv' -delete
tail -3 gpurun_out/prof_stream.log | cut -c1-300
