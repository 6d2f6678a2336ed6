#!/bin/bash
# round 4, Winograd route, first GPU call: parity tests of tests/test_gpu_wino.py, per-layer A/B (direct / planner / everywhere), joint bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4w1; mkdir -p $O
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_wino.py -q -s -p no:cacheprovider > $O/pytest_wino.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_wino.log
tail -5 $O/pytest_wino.log
for m in 0 2; do
  WINO=$m SKIP_WGRAD=1 ONLY="->" timeout 240 python tools/bench_conv.py 2>&1 | grep -E "layer|256|512|128->128|cls2 64|segmenter" > $O/conv_layers_wino$m.txt
done
paste -d'\n' $O/conv_layers_wino0.txt $O/conv_layers_wino2.txt | head -60
PNP_WINOGRAD=1 timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_joint_wino1.json 2> $O/bench_joint_wino1.err; tail -c 1500 $O/bench_joint_wino1.json
