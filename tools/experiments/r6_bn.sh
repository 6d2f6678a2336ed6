#!/bin/bash
# round 6: BN kernels after the thread-owns-a-channel-quad restructuring: parity (elementwise + teacher-forced B=2) and the joint step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_tfdoc_kats.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_teacher_forced.py -x -q -m gpu -k "not B16" 2>&1 | tail -3
for rep in 1 2; do python bench.py --no-cpu-baseline --no-sub --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
