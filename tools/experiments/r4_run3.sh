# round 4, GPU call 3: the resident kernels behind the autograd glue — parity tests, then the bf16 lines and a kernel table of the bf16 step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bf16r.py tests/test_gpu_bf16.py -q -s -x > $O/tests_bf16.log 2>&1; tail -5 $O/tests_bf16.log
grep "resident\|casts\|joint step bf16\|bf16 vs fp32" $O/tests_bf16.log | cut -c1-260
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; tail -1 $O/bench_bf16.json | cut -c1-1200
PNP_BF16R_OFF=1 timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --no-probe > $O/bench_bf16_off.json 2> $O/bench_bf16_off.err; tail -1 $O/bench_bf16_off.json | cut -c1-400
timeout 600 python bench.py --dtype bf16 --batch 32 --no-cpu-baseline --no-sub > $O/bench_bf16_B32.json 2>/dev/null; tail -1 $O/bench_bf16_B32.json | cut -c1-400
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub --dtype bf16"
timeout 420 rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o bf16 -- $B > $O/bench_prof_bf16.json 2>/dev/null
X=$(python -c "import json;r=json.loads(open('$O/bench_prof_bf16.json').read().strip().splitlines()[-1]);print(r['steps']*r['ms_per_step'])" 2>/dev/null)
python tools/rocpd_summary.py $(find $O/prof_bf16 -name "*.db" | head -1) $O/bf16_kernel_stats.txt --last-ms $X > /dev/null 2>&1
head -45 $O/bf16_kernel_stats.txt | cut -c1-175
rm -rf $O/prof_bf16
