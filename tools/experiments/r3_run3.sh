# round 3, GPU call 3: (1) MFMA dependency-chain micro-benchmark; (2) split-accumulator narrow tiles (variant libpnp_hip_sacc.so) — parity +
# per-layer A/B; (3) BN backward with the recomputed activation sign — parity + whole-step A/B; (4) n16 filter gradient vs ring kernel for
# 32->64 @ 256^2; (5) bf16 test with observed kernel names
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
P=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd
timeout 120 tools/experiments/bin/mfma_chain > $O/mfma_chain.txt 2>&1; cat $O/mfma_chain.txt
timeout 400 python -m pytest tests/test_gpu_conv.py tests/test_gpu_elementwise.py tests/test_gpu_bf16.py -x -q > $O/tests_main.log 2>&1; tail -3 $O/tests_main.log
PNP_LIB=$P/libpnp_hip_sacc.so timeout 300 python -m pytest tests/test_gpu_conv.py -x -q > $O/tests_conv_sacc.log 2>&1; tail -2 $O/tests_conv_sacc.log
timeout 200 python tools/bench_conv.py > $O/conv_layers_base.txt 2>&1
PNP_LIB=$P/libpnp_hip_sacc.so timeout 200 python tools/bench_conv.py > $O/conv_layers_sacc.txt 2>&1
PNP_N16W_MAXK=32 ONLY="32->64" timeout 100 python tools/bench_conv.py > $O/conv_layers_n16cap.txt 2>&1
paste -d'|' <(cut -c1-62 $O/conv_layers_base.txt) <(cut -c30-62 $O/conv_layers_sacc.txt) | grep -v amdgpu | head -40
grep "32->64" $O/conv_layers_base.txt $O/conv_layers_n16cap.txt
for v in "0 " "1 " "0 " "1 " "1 sacc"; do
  set -- $v
  L=$P/libpnp_hip${2:+_$2}.so
  PNP_LIB=$L PNP_BN_RECOMPUTE_SIGN=$1 timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resign=$1 lib=${2:-base}', r['value'], r['ms_per_step'], r['segmenter_step']['value'])" >> $O/ab.txt
done
cat $O/ab.txt
timeout 300 python -m pytest tests/test_gpu_teacher_forced_adv.py tests/test_gpu_teacher_forced.py -x -q -m "gpu and not slow" > $O/tests_tf.log 2>&1; tail -2 $O/tests_tf.log
