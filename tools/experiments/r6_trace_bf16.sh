#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6trace; mkdir -p $O
B="python bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-sub --graph off"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o bf16 -- $B > $O/bench_prof_bf16.json 2>/dev/null
X=$(python -c "import json;r=json.loads(open('$O/bench_prof_bf16.json').read().strip().splitlines()[-1]);print(r['steps']*r['ms_per_step'])" 2>/dev/null)
python tools/rocpd_summary.py $(find $O/prof_bf16 -name "*.db" | head -1) $O/bf16_kernel_stats.txt --last-ms $X > /dev/null 2>&1
rm -rf $O/prof_bf16
head -34 $O/bf16_kernel_stats.txt | cut -c1-60,100-175
