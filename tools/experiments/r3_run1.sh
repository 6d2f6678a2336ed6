# round 3, GPU call 1: the new adaptation-graph teacher-forced tests, the whole -m gpu suite after the host-side refactors, and A/B of two
# compile-time variants (libpnp_hip_occ3.so: 3 waves/SIMD for the 128x64 3x3 tiles; libpnp_hip_uni.so: scalar loader rows in the ring wgrad)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
P=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd
timeout 900 python -m pytest tests/test_gpu_teacher_forced_adv.py -x -q -s > $O/tf_adv.log 2>&1; tail -5 $O/tf_adv.log
timeout 1500 python -m pytest tests -m gpu -q --durations=12 --deselect tests/test_gpu_teacher_forced_adv.py > $O/tests.log 2>&1; tail -4 $O/tests.log
for v in "" occ3 uni; do
  L=$P/libpnp_hip${v:+_$v}.so
  PNP_LIB=$L timeout 200 python tools/bench_conv.py > $O/conv_layers_${v:-base}.txt 2>&1
done
PNP_LIB=$P/libpnp_hip_uni.so timeout 300 python -m pytest tests/test_gpu_conv.py -x -q > $O/tests_conv_uni.log 2>&1; tail -2 $O/tests_conv_uni.log
for v in "" occ3 uni ""; do
  L=$P/libpnp_hip${v:+_$v}.so
  PNP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${v:-base}', r['value'], r['ms_per_step'], r['segmenter_step']['value'])" >> $O/ab_variants.txt
done
cat $O/ab_variants.txt
paste -d'|' <(cut -c1-78 $O/conv_layers_base.txt) <(cut -c30-78 $O/conv_layers_occ3.txt) <(cut -c62-78 $O/conv_layers_uni.txt) | head -30
