# kernel trace of the one-utterance flow stage at a given generated-token count: bash tools/prof_flow_len.sh N [N ...]
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  d=$GRAFT_REPO_ROOT/gpurun_out/prof_flen_$n
  rm -rf $d
  LEN_CHILD=1 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/exp_flow_len.py $n > $d.log 2>&1
  (cd $GRAFT_REPO_ROOT && echo "== N = $n: $(grep 'frames' $d.log | tail -1)" && python tools/prof_summary.py $d 12 2>&1 | head -14)
  find $d -name '*_kernel_trace.csv' -delete
done
