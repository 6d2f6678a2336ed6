# round 3, GPU call 12: separable tap-validity masks in the tile set-up — parity + per-layer table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_conv.py -x -q > $O/tests_conv.log 2>&1; tail -2 $O/tests_conv.log
timeout 200 python tools/bench_conv.py 2>&1 | grep -v amdgpu > $O/conv_layers.txt; cat $O/conv_layers.txt
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('joint', r['value'], r['ms_per_step'], 'segmenter', r['segmenter_step']['value'])"
