#!/bin/bash
# round 5, call 8: persistent, cross-tile-pipelined GEMM of the route — parity, per-kernel per-layer A/B (PNP_WINO_PERSIST=0/1), joint-step A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5h; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_wino.py -q -x -p no:cacheprovider 2>&1 | tail -8) | tee $O/pytest_wino.log
L="g5/6,g7,g8,g10,cls2 128,cls3 256,cls5"
for p in 0 1; do echo "== PNP_WINO_PERSIST=$p"; PNP_WINO_PERSIST=$p PROF=1 SKIP_WGRAD=1 ONLY="$L" WINO=2 WINO_WGRAD=2 TILE=4 timeout 150 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | grep "gemm\|->" | tee $O/prof_persist$p.txt; done
run() { local tag=$1; shift
  env "$@" timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sub > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "
import json; r=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag joint', r['value'], r['ms_per_step'], r['roofline']['kernel'], r['roofline']['frac'], r['roofline']['avg_launch_ms'])" || tail -5 $O/bench_$tag.err
}
run off PNP_WINO_PERSIST=0
run on X=1
run off2 PNP_WINO_PERSIST=0
run on2 X=1
