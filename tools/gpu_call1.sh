cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/tests.log 2>&1; tail -40 $O/tests.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -4 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -3 $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o joint -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub > $O/bench_prof.json 2>$O/prof.err
python tools/rocpd_summary.py $(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1) $O/joint_kernel_stats.txt > /dev/null 2>&1; head -30 $O/joint_kernel_stats.txt
ls $O
