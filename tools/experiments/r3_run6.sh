# round 3, GPU call 6: ring filter gradient on 16^2 / 8^2 maps (parity + per-layer), gradient sinks through reshaped views (no aten::add left),
# bf16 joint-step test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
P=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_elementwise.py tests/test_gpu_bf16.py -x -q -s > $O/tests_kernels.log 2>&1; tail -2 $O/tests_kernels.log; grep -E "joint step bf16" $O/tests_kernels.log
timeout 600 python -m pytest tests/test_gpu_adversarial.py tests/test_gpu_teacher_forced_adv.py tests/test_gpu_segmenter.py -x -q -m "gpu and not slow" > $O/tests_steps.log 2>&1; grep -E "passed|failed" $O/tests_steps.log | tail -1
PNP_LIB=$P/libpnp_hip_prev.so ONLY=cls timeout 100 python tools/bench_conv.py > $O/conv_layers_prev.txt 2>&1
ONLY=cls timeout 100 python tools/bench_conv.py > $O/conv_layers_new.txt 2>&1
paste -d'|' <(cut -c1-18,63-80 $O/conv_layers_prev.txt) <(cut -c63-80 $O/conv_layers_new.txt) | grep -v amdgpu
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_joint -o joint -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub > $O/bench_prof_joint.json 2>/dev/null
X=$(python -c "import json;r=json.loads(open('$O/bench_prof_joint.json').read().strip().splitlines()[-1]);print(r['steps']*r['ms_per_step'])" 2>/dev/null)
python tools/rocpd_summary.py $(find $O/prof_joint -name "*.db" | head -1) $O/joint_kernel_stats.txt --last-ms $X > /dev/null 2>&1
grep -E "at::|Functor|elementwise_kernel|conv_wgrad_kernel|add_kernel|fill_kernel" $O/joint_kernel_stats.txt | cut -c1-170
rm -rf $O/prof_joint
