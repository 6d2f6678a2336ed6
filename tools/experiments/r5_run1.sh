#!/bin/bash
# round 5, call 1: F(4x4, 3x3) route — parity (both tiles), per-layer A/B direct / F(2x2) / F(4x4), joint + segmenter bench A/B by tile
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5a; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_wino.py -q -s -x -p no:cacheprovider 2>&1 | tail -60) > $O/pytest_wino.log; tail -5 $O/pytest_wino.log
(timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -8) | tee $O/smoke.log
L="g4 128,g5,g7,g8,g10,cls2 128,cls3,cls5"
for cfg in "0 2" "2 2" "2 4"; do set -- $cfg
  ONLY="$L" WINO=$1 WINO_WGRAD=$1 TILE=$2 timeout 120 python tools/bench_conv.py > $O/conv_layers_wino$1_tile$2.txt 2>&1
done
paste -d'\n' /dev/null; for f in $O/conv_layers_*.txt; do echo "== $f"; cat $f; done
for t in 4 2 4 2; do
  PNP_WINOGRAD_TILE=$t timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_tile${t}.json 2> $O/bench_tile${t}.err
  python -c "
import json; r=json.loads(open('$O/bench_tile${t}.json').read().strip().splitlines()[-1]); print('tile $t joint', r['value'], r['ms_per_step'], 'segmenter', r['segmenter_step']['value'], r['roofline']['kernel'], r['roofline']['frac'])" || tail -5 $O/bench_tile${t}.err
  cp gpurun_out/bench_kernels_joint_f32.json $O/bench_kernels_tile${t}.json 2>/dev/null
done
