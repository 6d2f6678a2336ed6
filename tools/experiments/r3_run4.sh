# round 3, GPU call 4: fragment lookahead of two slices in conv_taps3_kernel (A/B against libpnp_hip_prev.so = the build before it), n16/ring
# routing of the 32->64 @ 256^2 filter gradient
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
P=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd
timeout 300 python -m pytest tests/test_gpu_conv.py -x -q > $O/tests_conv.log 2>&1; tail -2 $O/tests_conv.log
PNP_LIB=$P/libpnp_hip_prev.so timeout 200 python tools/bench_conv.py > $O/conv_layers_prev.txt 2>&1
timeout 200 python tools/bench_conv.py > $O/conv_layers_new.txt 2>&1
paste -d'|' <(cut -c1-78 $O/conv_layers_prev.txt) <(cut -c30-78 $O/conv_layers_new.txt) | grep -v amdgpu | head -40
for v in prev "" prev ""; do
  L=$P/libpnp_hip${v:+_$v}.so
  PNP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${v:-new}', r['value'], r['ms_per_step'], r['segmenter_step']['value'])" >> $O/ab.txt
done
cat $O/ab.txt
