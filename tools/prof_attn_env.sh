# kernel trace of the flow stage alone on a 32-utterance batch, once per environment setting: bash tools/prof_attn_env.sh CV2_ATT_DMA8=0 CV2_ATT_DMA8=1 ...
cd /tmp && export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1)); d=$GRAFT_REPO_ROOT/gpurun_out/prof_atte_$i
  rm -rf $d
  env $e TAIL2_CHILD=1 CV2_FLOW_TAIL_ROWS2=2 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/exp_flow_tail2.py 32 > $d.log 2>&1
  (cd $GRAFT_REPO_ROOT && echo "== $e: $(grep 'flow ' $d.log | tail -1)" && python tools/prof_summary.py $d 2>&1 | grep "k_attn_est")
  find $d -name '*_kernel_trace.csv' -delete
done
