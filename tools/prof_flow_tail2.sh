# kernel trace of the flow stage alone on a 32-utterance batch: k_tail_rows<4> (CV2_FLOW_TAIL_ROWS2=0) against k_tail_rows2 with the chained QKV (2)
cd /tmp && export TMPDIR=/tmp
for m in 0 2; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_tail2_$m
  TAIL2_CHILD=1 CV2_FLOW_TAIL_ROWS2=$m rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tail2_$m -- python3 $GRAFT_REPO_ROOT/tools/exp_flow_tail2.py 32 > $GRAFT_REPO_ROOT/gpurun_out/prof_tail2_$m.log 2>&1
  (cd $GRAFT_REPO_ROOT && echo "== CV2_FLOW_TAIL_ROWS2=$m" && python tools/prof_summary.py gpurun_out/prof_tail2_$m 2>&1 | head -14)
  find $GRAFT_REPO_ROOT/gpurun_out/prof_tail2_$m -name '*_kernel_trace.csv' -delete
done
