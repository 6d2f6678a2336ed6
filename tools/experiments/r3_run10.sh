# round 3, GPU call 10: phased epilogue (loads / compute first, then stores only) — parity, stage trace, per-layer table, whole step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_elementwise.py tests/test_gpu_bf16.py -x -q > $O/tests_kernels.log 2>&1; tail -2 $O/tests_kernels.log
PNP_LIB=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd/libpnp_hip_trace.so timeout 200 python tools/experiments/stage_trace.py 2>&1 | grep -v amdgpu > $O/stage_trace.txt; cat $O/stage_trace.txt
timeout 200 python tools/bench_conv.py 2>&1 | grep -v amdgpu > $O/conv_layers.txt; cat $O/conv_layers.txt
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('joint', r['value'], r['ms_per_step'], 'segmenter', r['segmenter_step']['value'])" >> $O/bench.txt; done; cat $O/bench.txt
timeout 400 python -m pytest tests/test_gpu_teacher_forced_adv.py tests/test_gpu_teacher_forced.py tests/test_gpu_segmenter.py tests/test_gpu_adversarial.py -x -q -m "gpu and not slow" > $O/tests_steps.log 2>&1; grep -E "passed|failed" $O/tests_steps.log | tail -1
