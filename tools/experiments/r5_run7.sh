#!/bin/bash
# round 5, call 7: planner with the 64-channel layers on F(4x4) where measured to pay — parity, joint-step A/B against the previous planner (PNP_WINOGRAD4_LOW=60)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5g; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_wino.py -q -x -p no:cacheprovider 2>&1 | tail -4) | tee $O/pytest_wino.log
run() { local tag=$1; shift
  env "$@" timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "
import json; r=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag joint', r['value'], r['ms_per_step'], 'segmenter', r['segmenter_step']['value'], 'bf16', r.get('bf16_step',{}).get('value'), r['roofline']['kernel'], r['roofline']['frac'], r['roofline']['traffic'], len(json.dumps(r)))" || tail -5 $O/bench_$tag.err
}
run old PNP_WINOGRAD4_LOW=60
run new X=1
run old2 PNP_WINOGRAD4_LOW=60
run new2 X=1
run tile2 PNP_WINOGRAD_TILE=2
cp gpurun_out/bench_kernels_joint_f32.json $O/ 2>/dev/null
