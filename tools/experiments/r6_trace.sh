#!/bin/bash
# round 6: rocprofv3 kernel trace of the timed region of the joint / segmenter step -> per-symbol table
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6trace; mkdir -p $O
W=${1:-joint}
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub --graph off"
[ $W = seg ] && B="$B --workload segmenter"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$W -o $W -- $B > $O/bench_prof_$W.json 2>/dev/null
X=$(python -c "import json;r=json.loads(open('$O/bench_prof_$W.json').read().strip().splitlines()[-1]);print(r['steps']*r['ms_per_step'])" 2>/dev/null)
python tools/rocpd_summary.py $(find $O/prof_$W -name "*.db" | head -1) $O/${W}_kernel_stats.txt --last-ms $X > /dev/null 2>&1
rm -rf $O/prof_$W
head -60 $O/${W}_kernel_stats.txt | cut -c1-60,100-175
