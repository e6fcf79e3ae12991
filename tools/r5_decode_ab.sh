#!/bin/bash
# round 5: the one-row decode step's forms side by side on one box (CV2_STEP1: 0 = k_step, 1 = QA blocks, 2 = two-pair gate/up blocks, 3 = both)
cd "$(dirname "$0")/.."
R=$PWD
OUT=gpurun_out/r5_decode_ab.txt
: > $OUT
export CV2_MAXPOS=968
for m in ${MODES:-0 1 3 2 0 1 3}; do
  echo "== CV2_STEP1=$m" >> $OUT
  CV2_STEP1=$m python tools/bench_llm.py 1 400 255 2>&1 | grep "decode step" >> $OUT
done
for m in ${TL_MODES:-0 1}; do
  echo "== timeline CV2_STEP1=$m" >> $OUT
  CV2_STEP1=$m CV2_AMD_LIB=$R/cosyvoice2-eu_amd/cv2amd/libcv2amd_dbg.so python tools/dbg_chain.py 2>&1 | grep -v amdgpu.ids >> $OUT
done
cat $OUT
