# round 3, GPU call 2: three-stage narrow-tile kernel (conv_taps3_kernel, run-time switch PNP_CONV_TAPS3) + scalar-row ring wgrad as default:
# parity of every conv geometry, the adaptation-graph teacher-forced tests (B=2 and B=16), per-layer and whole-step A/B, train-gan end to end
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_conv.py tests/test_gpu_elementwise.py -x -q > $O/tests_conv.log 2>&1; tail -2 $O/tests_conv.log
timeout 900 python -m pytest tests/test_gpu_teacher_forced_adv.py -x -q -s > $O/tf_adv.log 2>&1; tail -3 $O/tf_adv.log
for t in 0 1; do
  PNP_CONV_TAPS3=$t timeout 200 python tools/bench_conv.py > $O/conv_layers_taps3_$t.txt 2>&1
done
paste -d'|' <(cut -c1-62 $O/conv_layers_taps3_0.txt) <(cut -c30-62 $O/conv_layers_taps3_1.txt) | head -30
for t in 0 1 0 1; do
  PNP_CONV_TAPS3=$t timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('taps3=$t', r['value'], r['ms_per_step'], r['segmenter_step']['value'])" >> $O/ab_taps3.txt
done
cat $O/ab_taps3.txt
timeout 400 python tools/e2e_gan.py 2>&1 | grep "E2E" > $O/e2e_gan.txt; cat $O/e2e_gan.txt
