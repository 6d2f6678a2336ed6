#!/bin/bash
# round 5, call 2: evidence on the F(4x4) tree — PMC passes (one counter set per pass; first on the single-layer microbench, then the joint step),
# rocprofv3 --kernel-trace --stats of the joint step, break-even layers under F(4x4), per-GPU batch sweep of the planner floors
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5b; mkdir -p $O
pmc() {   # pmc <outdir> <prefix> <secs> <counters...> -- <command...>
  local d=$1 o=$2 s=$3; shift 3; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -k 10 $s rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $d -o $o -- "$@" > $d.log 2>&1
  echo "PMC pass $d rc=$?"
}
SQ="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES"
export ONLY="g7/9 512" WINO=2 WINO_WGRAD=2 TILE=4
M="python tools/bench_conv.py"
pmc $O/pmcL_fetch f 150 FETCH_SIZE -- $M
pmc $O/pmcL_write w 150 WRITE_SIZE -- $M
pmc $O/pmcL_sq s 150 $SQ -- $M
python tools/pmc_summary.py $O/pmcL_fetch $O/pmcL_write $O/pmcL_sq $O/pmc_counters_layer512.json > /dev/null 2>$O/pmcL_summary.err; tail -2 $O/pmcL_summary.err
unset ONLY WINO WINO_WGRAD TILE
P1="python bench.py --steps 2 --warmup 1 --no-probe --no-cpu-baseline --no-sub --graph off"
pmc $O/pmc_fetch f 240 FETCH_SIZE -- $P1
if grep -q "rc=139\|Segmentation" $O/pmc_fetch.log 2>/dev/null || [ ! -d $O/pmc_fetch ]; then echo "retry without the filter-gradient side stream"; export PNP_WGRAD_STREAM=0; pmc $O/pmc_fetch f 240 FETCH_SIZE -- $P1; fi
pmc $O/pmc_write w 240 WRITE_SIZE -- $P1
pmc $O/pmc_sq s 240 $SQ -- $P1
unset PNP_WGRAD_STREAM
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_counters.json > /dev/null 2>$O/pmc_summary.err; tail -2 $O/pmc_summary.err
tail -3 $O/pmc_fetch.log
rm -rf $O/pmcL_fetch $O/pmcL_write $O/pmcL_sq $O/pmc_fetch $O/pmc_write $O/pmc_sq
# kernel trace of the joint step
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub --graph off"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_joint -o joint -- $B > $O/bench_prof_joint.json 2>$O/bench_prof_joint.err
X=$(python -c "import json;r=json.loads(open('$O/bench_prof_joint.json').read().strip().splitlines()[-1]);print(r['steps']*r['ms_per_step'])" 2>/dev/null)
python tools/rocpd_summary.py $(find $O/prof_joint -name "*.db" | head -1) $O/joint_kernel_stats.txt --last-ms $X > /dev/null 2>&1
head -30 $O/joint_kernel_stats.txt | cut -c1-150
rm -rf $O/prof_joint
# break-even of F(4x4) (mode 2 = wherever eligible) against the direct kernels (mode 0) on the narrow layers
L="g3 64,g4 64,g4 128,cls1 64->64,cls2 64,cls2 128,cls3 128"
for cfg in "0 4" "2 4"; do set -- $cfg
  ONLY="$L" WINO=$1 WINO_WGRAD=$1 TILE=$2 timeout 200 python tools/bench_conv.py > $O/narrow_layers_wino$1.txt 2>&1; cat $O/narrow_layers_wino$1.txt
done
# planner floors at small per-GPU batches: direct / F(2x2) / F(4x4) on the wide layers, B = 2, 4, 8
L="g5,g7,g8,g10,cls3 256,cls5"
for b in 2 4 8; do for cfg in "0 2" "2 2" "2 4"; do set -- $cfg
  echo "== B=$b WINO=$1 TILE=$2"; B=$b ONLY="$L" WINO=$1 WINO_WGRAD=$1 TILE=$2 timeout 100 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | tee $O/wide_B${b}_wino$1_tile$2.txt
done; done
