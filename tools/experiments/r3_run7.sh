# round 3, GPU call 7: persistent conv_taps3_kernel (workgroups stream over tiles; PNP_CONV_PERSIST=0/1) — parity, per-layer and whole-step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_elementwise.py -x -q > $O/tests_kernels.log 2>&1; tail -2 $O/tests_kernels.log
for t in 0 1; do PNP_CONV_PERSIST=$t ONLY=cls timeout 100 python tools/bench_conv.py > $O/conv_layers_persist_$t.txt 2>&1; done
paste -d'|' <(cut -c1-62 $O/conv_layers_persist_0.txt) <(cut -c30-62 $O/conv_layers_persist_1.txt) | grep -v amdgpu
for t in 0 1 0 1; do
  PNP_CONV_PERSIST=$t timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('persist=$t', r['value'], r['ms_per_step'], r['segmenter_step']['value'])" >> $O/ab.txt
done
cat $O/ab.txt
timeout 300 python -m pytest tests/test_gpu_teacher_forced_adv.py tests/test_gpu_segmenter.py -x -q -m "gpu and not slow" > $O/tests_steps.log 2>&1; grep -E "passed|failed" $O/tests_steps.log | tail -1
