#!/bin/bash
# round 5, call 11: entry-point loops end to end on the final tree (tfrecords -> feeder -> step), kernel trace of the bf16 joint step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5k; mkdir -p $O
timeout 300 python tools/e2e_segmenter.py 2>&1 | grep "E2E" | tee $O/e2e.txt
timeout 400 python tools/e2e_gan.py 2>&1 | grep "E2E" | tee $O/e2e_gan.txt
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub --graph off --dtype bf16"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o bf16 -- $B > $O/bench_prof_bf16.json 2>/dev/null
X=$(python -c "import json;r=json.loads(open('$O/bench_prof_bf16.json').read().strip().splitlines()[-1]);print(r['steps']*r['ms_per_step'])" 2>/dev/null)
python tools/rocpd_summary.py $(find $O/prof_bf16 -name "*.db" | head -1) $O/bf16_kernel_stats.txt --last-ms $X > /dev/null 2>&1
head -16 $O/bf16_kernel_stats.txt | cut -c1-160
rm -rf $O/prof_bf16
