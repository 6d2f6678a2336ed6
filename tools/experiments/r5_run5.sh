#!/bin/bash
# round 5, call 5: one-launch BN combine (tickets) + split-sum of the filter-gradient partials — parity, then A/B on the joint step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5e; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_wino.py tests/test_gpu_capture.py tests/test_gpu_dp.py -q -x -p no:cacheprovider 2>&1 | tail -15) > $O/pytest_subset.log; tail -4 $O/pytest_subset.log
ONLY="cls2 128,g5/6" PROF=1 WINO=2 WINO_WGRAD=2 TILE=4 timeout 100 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | grep "wgrad\|layer\|->"
run() { local tag=$1; shift
  env "$@" timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "
import json; r=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag joint', r['value'], r['ms_per_step'], 'segmenter', r['segmenter_step']['value'], 'bf16', r.get('bf16_step',{}).get('value'), r['roofline']['kernel'], r['roofline']['frac'], r['roofline']['traffic'], len(json.dumps(r)))" || tail -5 $O/bench_$tag.err
}
run multi PNP_BN_ONE_LAUNCH=0
run one X=1
run multi2 PNP_BN_ONE_LAUNCH=0
run one2 X=1
cp gpurun_out/bench_kernels_joint_f32.json $O/ 2>/dev/null
# launches per step (kernel trace of 3 steps)
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_joint -o joint -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-sub --graph off > $O/bench_prof_joint.json 2>$O/bench_prof_joint.err
X=$(python -c "import json;r=json.loads(open('$O/bench_prof_joint.json').read().strip().splitlines()[-1]);print(r['steps']*r['ms_per_step'])" 2>/dev/null)
python tools/rocpd_summary.py $(find $O/prof_joint -name "*.db" | head -1) $O/joint_kernel_stats.txt --last-ms $X > /dev/null 2>&1
head -24 $O/joint_kernel_stats.txt | cut -c1-150
rm -rf $O/prof_joint
