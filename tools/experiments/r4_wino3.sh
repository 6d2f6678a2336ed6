#!/bin/bash
# round 4, Winograd route on by default: kernel trace of the joint step's timed region, then the 20-step bench line (no CPU baseline:
# unchanged since profiles/r04_bench_n1.json), then the segmenter line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4w3; mkdir -p $O
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub --graph off"
timeout 75 rocprofv3 --kernel-trace --stats -d $O/prof_joint -o joint -- $B > $O/bench_prof_joint.json 2>/dev/null
X=$(python -c "import json;r=json.loads(open('$O/bench_prof_joint.json').read().strip().splitlines()[-1]);print(r['steps']*r['ms_per_step'])" 2>/dev/null)
timeout 30 python tools/rocpd_summary.py $(find $O/prof_joint -name "*.db" | head -1) $O/joint_kernel_stats.txt --last-ms $X > /dev/null 2>&1
rm -rf $O/prof_joint
head -8 $O/joint_kernel_stats.txt | cut -c1-170
timeout 40 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; cp gpurun_out/bench_kernels_joint_f32.json $O/ 2>/dev/null
tail -c 400 $O/bench_n1.json
