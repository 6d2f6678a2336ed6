cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r2j && mkdir -p $O
for w in joint segmenter; do
rocprofv3 --kernel-trace --stats -d $O/prof_$w -o $w -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub --workload $w > $O/bench_prof_$w.json 2>/dev/null
X=$(python -c "import json;r=json.loads(open('$O/bench_prof_$w.json').read().strip().splitlines()[-1]);print(r['steps']*r['ms_per_step'])")
python tools/rocpd_summary.py $(find $O/prof_$w -name "*.db" | head -1) $O/${w}_kernel_stats.txt --last-ms $X > /dev/null 2>$O/err_$w.txt; cat $O/err_$w.txt | tail -2
rm -rf $O/prof_$w
done
head -60 $O/joint_kernel_stats.txt | cut -c1-100,113-175
