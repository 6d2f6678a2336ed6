# round 3, GPU call 5: BN-backward sums from the data-gradient epilogue (parity + A/B), fan-out sums / fills through libpnp_hip.so
# (whole-step parity tests + a kernel trace to count what torch still launches)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_conv.py -x -q -s > $O/tests_kernels.log 2>&1; tail -2 $O/tests_kernels.log; grep -E "^bnred|residual block" $O/tests_kernels.log | head -12
timeout 900 python -m pytest tests/test_gpu_segmenter.py tests/test_gpu_adversarial.py tests/test_gpu_dp.py tests/test_gpu_entrypoints.py tests/test_gpu_golden.py -x -q -m "gpu and not slow" > $O/tests_steps.log 2>&1; tail -3 $O/tests_steps.log
for v in 0 1 0 1; do
  PNP_BN_BWD_FROM_DGRAD=$v timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bnred=$v', r['value'], r['ms_per_step'], r['segmenter_step']['value'])" >> $O/ab.txt
done
cat $O/ab.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_joint -o joint -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub > $O/bench_prof_joint.json 2>/dev/null
X=$(python -c "import json;r=json.loads(open('$O/bench_prof_joint.json').read().strip().splitlines()[-1]);print(r['steps']*r['ms_per_step'])" 2>/dev/null)
python tools/rocpd_summary.py $(find $O/prof_joint -name "*.db" | head -1) $O/joint_kernel_stats.txt --last-ms $X > /dev/null 2>&1
head -24 $O/joint_kernel_stats.txt | cut -c1-170; grep -E "at::|Functor|elementwise_kernel" $O/joint_kernel_stats.txt | cut -c1-170
rm -rf $O/prof_joint
