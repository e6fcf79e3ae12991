# kernel trace of one prompt's prefill with and without the split-K down projection (tools/exp_prefill1.py)
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_pf$m
  CV2_PREFILL_SPLITK=$m rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_pf$m -- python3 $GRAFT_REPO_ROOT/tools/exp_prefill1.py > $GRAFT_REPO_ROOT/gpurun_out/prof_pf$m.log 2>&1
  echo "== CV2_PREFILL_SPLITK=$m"
  (cd $GRAFT_REPO_ROOT && python tools/prof_summary.py gpurun_out/prof_pf$m 2>&1 | head -9)
  find $GRAFT_REPO_ROOT/gpurun_out/prof_pf$m -name '*_kernel_trace.csv' -delete
done
