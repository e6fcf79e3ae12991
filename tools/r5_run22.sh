for r in 1 2; do for l in libcv2amd_base.so libcv2amd.so libcv2amd_b6.so; do echo "== $l"; CV2_AMD_LIB=$PWD/cosyvoice2-eu_amd/cv2amd/$l python tools/bench_hift.py 500 30 2>&1 | tail -1; done; done
