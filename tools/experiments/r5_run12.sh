#!/bin/bash
# round 5, call 12: tile ORDER of the persistent GEMM (PNP_WINO_GN; results bit-identical): per-layer GEMM times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5l; mkdir -p $O
L="g7,g8,g10,cls3 256"
for gn in 4 0 1 2 8; do echo "== PNP_WINO_GN=$gn"; PNP_WINO_GN=$gn PROF=1 SKIP_WGRAD=1 ONLY="$L" WINO=2 WINO_WGRAD=2 TILE=4 timeout 100 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | grep "wino_gemm" | tee $O/gn$gn.txt; done
