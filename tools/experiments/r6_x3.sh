#!/bin/bash
# round 6: split-bf16 GEMM of the Winograd route — parity of the route in both arithmetic modes, then per-layer / per-kernel timing
# usage: r6_x3.sh <tag> [layers]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-a}; O=gpurun_out/r6x3_$TAG; mkdir -p $O
LAYERS=${2:-"g5/6,g7,g8,g10,cls2 128,cls3 256,cls5,cls1 64,cls2 64"}
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests/test_gpu_wino.py -x -q -m gpu -k "fwd_dgrad or epilogues" -s 2>&1 | grep -v amdgpu.ids | tail -80 > $O/pytest.txt
tail -5 $O/pytest.txt
fi
for x3 in 0 1; do
  echo "== X3=$x3" | tee -a $O/layers.txt
  X3=$x3 PROF=1 ONLY="$LAYERS" WINO=1 WINO_WGRAD=1 TILE=4 timeout 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | tee -a $O/layers.txt | grep -E "^\S|wino_gemm|wino_in|wino_out" | grep -v wgrad
done
