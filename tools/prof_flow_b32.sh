# kernel trace of the flow stage alone on a 32-utterance batch, estimator attention with and without LDS DMA staging (tools/exp_flow_b32.py)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_flow32
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_flow32 -- python3 $GRAFT_REPO_ROOT/tools/exp_flow_b32.py > $GRAFT_REPO_ROOT/gpurun_out/prof_flow32.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py gpurun_out/prof_flow32 2>&1 | head -24
find gpurun_out/prof_flow32 -name '*_kernel_trace.csv' -delete
tail -3 gpurun_out/prof_flow32.log
