#!/bin/bash
# Round 6 (the round-5 review's item 4a): what would k_tail_rows2's feed-forward phase gain from 32 x 32 MFMA tiles?  TIMING-ONLY diagnostic build:
#   ./build.sh -DTR2_DIAG_MFMA32 -o cosyvoice2-eu_amd/cv2amd/libcv2amd_mfma32.so
# replaces the eight v_mfma_f32_16x16x32_bf16 of every k-step of the batch tail kernel by four v_mfma_f32_32x32x16_bf16 on the same registers
# (numerically meaningless, same matrix-core time and operand traffic, half the vector-issue cycles taken from the GELU beside them).
# Kernel trace of the flow stage alone on a 32-utterance batch, product build against the diagnostic one.   bash tools/exp_tail_mfma32.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in libcv2amd.so libcv2amd_mfma32.so; do
  rm -rf $R/gpurun_out/prof_mfma32
  CV2_AMD_LIB=$R/cosyvoice2-eu_amd/cv2amd/$lib TAIL2_CHILD=1 CV2_FLOW_TAIL_ROWS2=2 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_mfma32 -- python3 $R/tools/exp_flow_tail2.py 32 > $R/gpurun_out/prof_mfma32.log 2>&1
  (cd $R && echo "== $lib" && tail -1 gpurun_out/prof_mfma32.log && python tools/prof_summary.py gpurun_out/prof_mfma32 2>&1 | head -8)
done
rm -rf $R/gpurun_out/prof_mfma32
