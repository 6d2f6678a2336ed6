#!/bin/bash
# round 6: what bounds wino_gemm_x3_kernel — timing-only ablations of the 128x128 instance (variant build -DPNP_X3_ABLATIONS):
# PNP_X3_ABL=1 no MFMA / 2 no LDS-DMA in the main loop / 3 no fragment reads (results are garbage by construction)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6abl; mkdir -p $O
for abl in ${ABLS:-0 1 2 3}; do
  echo "== PNP_X3_ABL=$abl" | tee -a $O/abl.txt
  PNP_LIB=$PWD/medical-cross-modality-domain-adaptation_amd/libpnp_hip_abl.so PNP_X3_ABL=$abl X3=1 PROF=1 SKIP_WGRAD=1 ONLY="g5/6,g7/9,g10" WINO=1 TILE=4 timeout 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | grep -E "fwd +wino_gemm" | tee -a $O/abl.txt
done
