#!/bin/bash
# Build libcv2amd.so for gfx950 in-tree (the .so travels with gpurun snapshots; it is git-ignored).
set -e
cd "$(dirname "$0")"
SRC=cosyvoice2-eu_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form=1 $SRC/*.hip -o cosyvoice2-eu_amd/cv2amd/libcv2amd.so "$@"
