#!/bin/bash
# round 4, Winograd filter gradient (PNP_WINOGRAD_WGRAD): parity tests, per-layer A/B, segmenter + joint bench lines with it on
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4w6; mkdir -p $O
timeout 40 python -m pytest tests/test_gpu_wino.py -q -s -p no:cacheprovider > $O/pytest_wino.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_wino.log
grep -E "passed|failed|FAILED|Error" $O/pytest_wino.log | tail -8
for m in 0 2; do
  WINO=1 WINO_WGRAD=$m ONLY="->" timeout 25 python tools/bench_conv.py 2>&1 | grep -E "layer|128->256|256|512|segmenter" > $O/conv_layers_wgrad$m.txt
done
paste -d'\n' $O/conv_layers_wgrad0.txt $O/conv_layers_wgrad2.txt | cut -c1-150
PNP_WINOGRAD_WGRAD=1 timeout 30 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_joint_wgrad1.json 2> $O/bench_joint_wgrad1.err
python -c "
import json; r=json.loads(open('$O/bench_joint_wgrad1.json').read().strip().splitlines()[-1]); print('joint', r['value'], r['ms_per_step'], 'segmenter', r['segmenter_step']['value'])"
