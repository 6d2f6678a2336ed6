#!/bin/bash
# round 4, final tree (Winograd route on for forward / data gradient / filter gradient): smoke, route + capture tests, the 20-step bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4w7; mkdir -p $O
timeout 25 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; cp gpurun_out/bench_kernels_joint_f32.json $O/ 2>/dev/null
python -c "
import json; r=json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1]); print('joint', r['value'], r['ms_per_step'], 'segmenter', r['segmenter_step']['value'], r['roofline']['kernel'], r['roofline']['frac'])"
(timeout 25 python __graft_entry__.py --smoke 2>&1 | tail -4) | tee $O/smoke.log
timeout 40 python -m pytest tests/test_gpu_wino.py tests/test_gpu_capture.py -q -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest_wino_capture.log
