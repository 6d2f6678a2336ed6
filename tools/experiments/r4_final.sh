# round 4, final-tree collection (the -m gpu suite ran in its own call): bench lines, kernel traces of the timed regions, per-layer
# tables, PMC passes of the bf16 symbols (the fp32 symbols did not change since profiles/r04_pmc_counters.json)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4z; mkdir -p $O
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 300 $O/bench_n1.json; cp gpurun_out/bench_kernels_joint_f32.json $O/ 2>/dev/null
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16_n1.json 2>/dev/null; cp gpurun_out/bench_kernels_joint_bf16.json $O/ 2>/dev/null
timeout 600 python bench.py --dtype bf16 --batch 32 --no-cpu-baseline --no-sub > $O/bench_bf16_B32_n1.json 2>/dev/null
timeout 900 python bench.py --workload segmenter --no-sub --cpu-small-batch 0 > $O/bench_segmenter_n1.json 2>/dev/null
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub --graph off"
timeout 420 rocprofv3 --kernel-trace --stats -d $O/prof_joint -o joint -- $B > $O/bench_prof_joint.json 2>/dev/null
timeout 420 rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o bf16 -- $B --dtype bf16 > $O/bench_prof_bf16.json 2>/dev/null
for w in joint bf16; do
  X=$(python -c "import json;r=json.loads(open('$O/bench_prof_$w.json').read().strip().splitlines()[-1]);print(r['steps']*r['ms_per_step'])" 2>/dev/null)
  python tools/rocpd_summary.py $(find $O/prof_$w -name "*.db" | head -1) $O/${w}_kernel_stats.txt --last-ms $X > /dev/null 2>&1
done
rm -rf $O/prof_joint $O/prof_bf16
timeout 400 python tools/bench_bf16r.py > $O/conv_layers_bf16r.txt 2>/dev/null
PMC_OK=1
pmc() {
  [ "$PMC_OK" = 1 ] || return 0
  local d=$1 o=$2; shift 2; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -k 10 240 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $d -o $o -- "$@" > /dev/null 2>&1
  local rc=$?; if [ $rc -ge 124 ]; then echo "PMC pass $d timed out (rc $rc): skipping the remaining passes"; PMC_OK=0; fi
}
PB="python bench.py --steps 2 --warmup 1 --no-probe --no-cpu-baseline --no-sub --graph off --dtype bf16"
pmc $O/pmc_bf16_fetch f FETCH_SIZE -- $PB
pmc $O/pmc_bf16_write w WRITE_SIZE -- $PB
pmc $O/pmc_bf16_sq s GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES -- $PB
python tools/pmc_summary.py $O/pmc_bf16_fetch $O/pmc_bf16_write $O/pmc_bf16_sq $O/bf16_pmc_counters.json > /dev/null 2>$O/pmc_summary.err
rm -rf $O/pmc_bf16_fetch $O/pmc_bf16_write $O/pmc_bf16_sq
for f in bench_n1 bench_bf16_n1 bench_bf16_B32_n1 bench_segmenter_n1; do echo "$f: $(tail -1 $O/$f.json | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"], len(json.dumps(r)))')"; done
du -sh $O; ls $O
