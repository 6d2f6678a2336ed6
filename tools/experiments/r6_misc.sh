#!/bin/bash
# round 6: (1) two-level split-K reduction on/off within one run; (2) the bf16 joint step eager vs captured at B = 16
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -2
bash tools/experiments/r6_step.sh PNP_SPLITK_TWO_LEVEL "0 1"
for g in off on; do echo "== bf16 --graph $g"; timeout 300 python bench.py --dtype bf16 --graph $g --no-cpu-baseline --no-sub --no-probe --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('step_capture'))"; done
