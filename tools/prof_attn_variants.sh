# kernel trace of the flow stage alone on a 32-utterance batch, once per library variant (libcv2amd_<v>.so): the estimator attention's
# average launch time.  bash tools/prof_attn_variants.sh att0 att1 ...   (the stage's mels are garbage for the diagnostic builds)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_attv_$v
  CV2_AMD_LIB=$GRAFT_REPO_ROOT/cosyvoice2-eu_amd/cv2amd/libcv2amd_$v.so TAIL2_CHILD=1 CV2_FLOW_TAIL_ROWS2=2 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_attv_$v -- python3 $GRAFT_REPO_ROOT/tools/exp_flow_tail2.py 32 > $GRAFT_REPO_ROOT/gpurun_out/prof_attv_$v.log 2>&1
  (cd $GRAFT_REPO_ROOT && echo "== $v: $(grep 'flow ' gpurun_out/prof_attv_$v.log | tail -1)" && python tools/prof_summary.py gpurun_out/prof_attv_$v 2>&1 | grep "k_attn_est\|k_tail_rows2<true>")
  find $GRAFT_REPO_ROOT/gpurun_out/prof_attv_$v -name '*_kernel_trace.csv' -delete
done
