#!/bin/bash
# round 5, call 6: 128x64 GEMM tile of the route for <= 64-column GEMMs — parity, then the narrow layers with the planner's break-even lowered
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5f; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_wino.py -q -x -p no:cacheprovider 2>&1 | tail -6) | tee $O/pytest_wino.log
L="g3 32,g3 64,g4 64,g4 128,cls1 32,cls1 64->64,cls2 64,cls2 128"
echo "== direct"; ONLY="$L" WINO=0 WINO_WGRAD=0 timeout 200 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | tee $O/narrow_direct.txt
echo "== F(4x4) forced (mode 2), 128x64 tiles for K <= 64"; PROF=1 ONLY="$L" WINO=2 WINO_WGRAD=2 TILE=4 timeout 200 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | tee $O/narrow_f4.txt
(timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -4) | tee $O/smoke.log
(timeout 500 python -m pytest tests/test_gpu_teacher_forced_adv.py -q -s -p no:cacheprovider -k B16 2>&1 | grep -E "whole step|passed|failed|^E ") | tee $O/pytest_tf_adv_B16.log
