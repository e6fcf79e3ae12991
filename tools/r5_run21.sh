for r in 1 2; do
for l in libcv2amd_ab.so libcv2amd.so; do
  echo "== $l"; TAIL2_CHILD=1 CV2_FLOW_TAIL_ROWS2=2 CV2_AMD_LIB=$PWD/cosyvoice2-eu_amd/cv2amd/$l python tools/exp_flow_tail2.py 1 2>&1 | grep "flow"
done; done
