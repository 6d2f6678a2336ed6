#!/bin/bash
# round 4, final tree: kernel trace of the joint step's timed region (the last 20 GPU-seconds of the round)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4w8; mkdir -p $O
timeout 14 rocprofv3 --kernel-trace --stats -d $O/prof_joint -o joint -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub --graph off > $O/bench_prof_joint.json 2>/dev/null
X=$(python -c "import json;r=json.loads(open('$O/bench_prof_joint.json').read().strip().splitlines()[-1]);print(r['steps']*r['ms_per_step'])" 2>/dev/null)
timeout 6 python tools/rocpd_summary.py $(find $O/prof_joint -name "*.db" | head -1) $O/joint_kernel_stats.txt --last-ms $X > /dev/null 2>&1
rm -rf $O/prof_joint
head -6 $O/joint_kernel_stats.txt | cut -c1-150
