# round 3, GPU call 14: (a) why the ASAN GPU run of collection PART B died after test_abi's 7 dots (exit code, stderr, -v);
# (b) 16-channel layers on the fp32 16x16x4 kernels in bf16 mode too (PNP_N16_F32ONLY=1 = round 2's routing): tests + A/B;
# (c) the dis / gen split of the joint step (tools/bench_gan.py) for round 3's kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n; mkdir -p $O
P=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd
timeout 400 python -m pytest tests/test_gpu_bf16.py -x -q -s > $O/tests_bf16.log 2>&1; tail -3 $O/tests_bf16.log
one() { timeout 300 python bench.py --dtype bf16 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 bf16 joint', r['value'], r['ms_per_step'], 'segmenter', r['segmenter_step']['value'])"; }
PNP_N16_F32ONLY=1 one old; one new; PNP_N16_F32ONLY=1 one old; one new
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16_n1.json 2>/dev/null
timeout 300 python tools/bench_gan.py 2>/dev/null | tail -1 > $O/gan_steps_B16.json; cat $O/gan_steps_B16.json
DTYPE=bf16 timeout 300 python tools/bench_conv.py > $O/conv_layers_bf16.txt 2>/dev/null; head -8 $O/conv_layers_bf16.txt
ASAN_LIB=$(/opt/rocm/bin/hipcc -print-file-name=libclang_rt.asan-x86_64.so)
for opts in "detect_leaks=0:protect_shadow_gap=0" "detect_leaks=0:protect_shadow_gap=0:handle_segv=0:allow_user_segv_handler=1"; do
  echo "== ASAN_OPTIONS=$opts" >> $O/asan_gpu.log
  LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=$opts PNP_LIB=$P/libpnp_hip_asan.so HSA_XNACK=0 \
    timeout 300 python -X faulthandler -m pytest tests/test_abi.py tests/test_gpu_conv.py -q -x -v -k "abi or (3 and 16)" >> $O/asan_gpu.log 2>&1
  echo "rc=$?" >> $O/asan_gpu.log
done
tail -30 $O/asan_gpu.log | cut -c1-220
