#!/bin/bash
# round 5, call 10: final check of the persistent GEMM tree — route tests (incl. the persistent / tail-split workers), capture, bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5j; mkdir -p $O
(timeout 500 python -m pytest tests/test_gpu_wino.py tests/test_gpu_capture.py tests/test_gpu_conv.py -q -x -s -p no:cacheprovider 2>&1 | grep -E "persistent GEMM|recycled|joint steps|passed|failed|Error" | tail -30) | tee $O/pytest.log
timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "
import json; r=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('joint', r['value'], r['ms_per_step'], 'segmenter', r['segmenter_step']['value'], 'bf16', r.get('bf16_step',{}).get('value'), r['roofline']['kernel'], r['roofline']['frac'], len(json.dumps(r)))" || tail -5 $O/bench.err
