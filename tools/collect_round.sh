# Regenerates everything under profiles/ for the current round on a GPU box:  bash tools/collect_round.sh   (raw output: gpurun_out/$R/)
# Every step runs under its own `timeout`: in round 2 a `rocprofv3 --pmc` pass hung and ate the remaining 26 GPU-minutes of the round.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${R:-r2f}; O=gpurun_out/$R; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/tests.log 2>&1; tail -3 $O/tests.log; fi
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
# the driver's line (joint segmenter+GAN step, segmenter sub-record, joint cpu_baseline), the segmenter workload as its own line, bf16
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 400 $O/bench_n1.json
timeout 600 python bench.py --workload segmenter --no-sub > $O/bench_segmenter_n1.json 2>/dev/null
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16_n1.json 2>/dev/null
# kernel traces (rocprofv3 --kernel-trace --stats), same command lines as the bench
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub"
timeout 420 rocprofv3 --kernel-trace --stats -d $O/prof_joint -o joint -- $B > $O/bench_prof_joint.json 2>/dev/null
timeout 420 rocprofv3 --kernel-trace --stats -d $O/prof_seg -o seg -- $B --workload segmenter > $O/bench_prof_seg.json 2>/dev/null
[ -z "$FAST" ] && timeout 420 rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o bf16 -- $B --dtype bf16 > $O/bench_prof_bf16.json 2>/dev/null
# per-kernel tables of the TIMED region only (warm-up and the joint workload's BN calibration forwards come before it): the dispatches that
# start within the last steps * ms_per_step milliseconds of the trace
for w in joint seg bf16; do
  [ -f $O/bench_prof_$w.json ] || continue
  X=$(python -c "import json;r=json.loads(open('$O/bench_prof_$w.json').read().strip().splitlines()[-1]);print(r['steps']*r['ms_per_step'])" 2>/dev/null)
  python tools/rocpd_summary.py $(find $O/prof_$w -name "*.db" | head -1) $O/${w}_kernel_stats.txt --last-ms $X > /dev/null 2>&1
done
head -12 $O/joint_kernel_stats.txt | cut -c1-170
# PMC passes, each on its own (never together with other trace domains)
P="python bench.py --steps 2 --warmup 1 --no-probe --no-cpu-baseline --no-sub"
timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- $P > /dev/null 2>&1
timeout 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- $P > /dev/null 2>&1
timeout 420 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o s -- $P > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_counters.json > /dev/null 2>$O/pmc_summary.err; tail -2 $O/pmc_summary.err
# instruction mix of the filter-gradient kernel (segmenter workload): separate small passes, a pass with an unknown counter name just fails
PS="python bench.py --workload segmenter --steps 2 --warmup 1 --no-probe --no-cpu-baseline --no-sub"
timeout 420 rocprofv3 -L > $O/counters_available.txt 2>&1
timeout 420 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_seg_sq -o s -- $PS > /dev/null 2>&1
timeout 420 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_seg_insts -o i -- $PS > /dev/null 2>&1
timeout 420 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/pmc_seg_insts2 -o j -- $PS > /dev/null 2>&1
python tools/pmc_raw.py $O/pmc_seg_sq $O/pmc_seg_insts $O/pmc_seg_insts2 > $O/pmc_segmenter_raw.txt 2>&1; head -30 $O/pmc_segmenter_raw.txt
# per-layer tables
timeout 300 python tools/bench_conv.py > $O/conv_layers_f32.txt 2>/dev/null
DTYPE=bf16 timeout 300 python tools/bench_conv.py > $O/conv_layers_bf16.txt 2>/dev/null
# A/B: BN statistics from the conv epilogue vs the reduction pass
if [ -z "$FAST" ]; then
for i in 1 2; do for f in 1 0; do PNP_FUSE_BN_STATS=$f python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FUSE_BN_STATS=$f', r['value'], r['ms_per_step'], r['segmenter_step']['value'])" >> $O/ab_bn_stats.txt; done; done; cat $O/ab_bn_stats.txt
python tools/e2e_segmenter.py 2>&1 | grep "E2E" > $O/e2e.txt; cat $O/e2e.txt
fi
rm -rf $O/prof_joint $O/prof_seg $O/prof_bf16     # the sqlite traces are large; the summaries stay
ls $O
