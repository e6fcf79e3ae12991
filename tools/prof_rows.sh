# kernel trace of decode steps at $1 live rows
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_rows$1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_rows$1 -- python3 $GRAFT_REPO_ROOT/tools/prof_rows.py $1 16 > $GRAFT_REPO_ROOT/gpurun_out/prof_rows$1.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py gpurun_out/prof_rows$1 2>&1 | head -${2:-16}
find gpurun_out/prof_rows$1 -name '*_kernel_trace.csv' -delete
