# Regenerates everything under profiles/ for the current round on a GPU box:  bash tools/collect_round.sh   (raw output: gpurun_out/$R/)
# Every step runs under its own `timeout`; the benchmark lines come FIRST (a box that has just run the 12-minute test suite clocks ~3 % lower),
# the `rocprofv3 --pmc` passes LAST and guarded: in round 2 one hung pass ate the remaining 26 GPU-minutes of the round.
# PART=A: smoke, bench lines, kernel traces, per-layer tables, entry-point loops, the whole -m gpu suite.  PART=B: UBSan run, CPU path at
# B=16, PMC passes.  Default: both.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${R:-r4z}; O=gpurun_out/$R; mkdir -p $O
P=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd
PART=${PART:-AB}
if [[ $PART == *A* ]]; then
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
# the driver's line (joint segmenter+GAN step, segmenter sub-record, joint cpu_baseline), the segmenter workload as its own line, bf16
timeout 1500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 400 $O/bench_n1.json
timeout 600 python bench.py --workload segmenter --no-sub > $O/bench_segmenter_n1.json 2>/dev/null
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16_n1.json 2>/dev/null
[ -z "$FAST" ] && timeout 600 python bench.py --dtype bf16 --batch 32 --no-cpu-baseline --no-sub > $O/bench_bf16_B32_n1.json 2>/dev/null
# kernel traces (rocprofv3 --kernel-trace --stats), same command lines as the bench
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub --graph off"
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
# per-layer tables
timeout 300 python tools/bench_conv.py > $O/conv_layers_f32.txt 2>/dev/null
DTYPE=bf16 timeout 300 python tools/bench_conv.py > $O/conv_layers_bf16.txt 2>/dev/null
timeout 400 python tools/bench_bf16r.py > $O/conv_layers_bf16r.txt 2>/dev/null       # the bf16-RESIDENT kernels next to the staged-rounding ones
if [ -z "$FAST" ]; then
python tools/e2e_segmenter.py 2>&1 | grep "E2E" > $O/e2e.txt; cat $O/e2e.txt
timeout 400 python tools/e2e_gan.py 2>&1 | grep "E2E" > $O/e2e_gan.txt; cat $O/e2e_gan.txt
fi
# -s: the parity tests PRINT their measured errors (the whole-step bars are set from these numbers: profiles/rNN_pytest_gpu.log)
if [ -z "$SKIP_TESTS" ]; then timeout 1800 python -m pytest tests -m gpu -q -s --durations=15 > $O/tests.log 2>&1; tail -3 $O/tests.log; fi
rm -rf $O/prof_joint $O/prof_seg $O/prof_bf16     # the sqlite traces are large; the summaries stay
fi
if [[ $PART == *B* ]]; then
# (PART B alone starts on a fresh box: the driver's line once more, as the driver itself measures it)
[[ $PART == *A* ]] || { timeout 1500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 300 $O/bench_n1.json; }
# host side of the library under UBSan + libstdc++ assertions (`make ubsan`; device code uninstrumented).  ASAN cannot run on the GPU box:
# ROCm's ASAN runtime intercepts hsa_amd_memory_pool_allocate and this image ships no ASAN ROCr (profiles/r03_asan_gpu.log)
if [ -f $P/libpnp_hip_ubsan.so ]; then
  PNP_LIB=$P/libpnp_hip_ubsan.so UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 timeout 900 python -m pytest tests/test_abi.py tests/test_api_errors.py \
    tests/test_gpu_conv.py tests/test_gpu_elementwise.py tests/test_gpu_loss_optim.py tests/test_gpu_adversarial.py tests/test_gpu_bf16.py -q > $O/ubsan_gpu.log 2>&1
  echo "rc=$? runtime-error lines: $(grep -c 'runtime error' $O/ubsan_gpu.log)" >> $O/ubsan_gpu.log; tail -3 $O/ubsan_gpu.log
fi
# (the CPU path at the GPU line's own batch is part of the driver's line since round 4: cpu_baseline runs B=16, 1 warm-up + 3 timed steps)
# ---- PMC passes LAST, each on its own (never together with other trace domains), each under a short timeout; after the first pass that
# times out the rest are skipped (round 2 lost 26 GPU-minutes to one hung pass)
PMC_OK=1
pmc() {   # pmc <outdir> <prefix> <counters...> -- <command...>
  [ "$PMC_OK" = 1 ] || return 0
  local d=$1 o=$2; shift 2; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -k 10 240 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $d -o $o -- "$@" > /dev/null 2>&1
  local rc=$?; if [ $rc -ge 124 ]; then echo "PMC pass $d timed out (rc $rc): skipping the remaining passes"; PMC_OK=0; fi
}
P1="python bench.py --steps 2 --warmup 1 --no-probe --no-cpu-baseline --no-sub --graph off"
pmc $O/pmc_fetch f FETCH_SIZE -- $P1
pmc $O/pmc_write w WRITE_SIZE -- $P1
pmc $O/pmc_sq s GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES -- $P1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_counters.json > /dev/null 2>$O/pmc_summary.err; tail -2 $O/pmc_summary.err
# bf16 symbols (configs[4]): HBM traffic + matrix-pipe utilisation (B=32: the configs[4] per-GPU batch — at B=16 the 256x128 tiles of
# the 32^2 layers are exactly one dispatch round)
PB="$P1 --dtype bf16 --graph off"
pmc $O/pmc_bf16_fetch f FETCH_SIZE -- $PB
pmc $O/pmc_bf16_write w WRITE_SIZE -- $PB
pmc $O/pmc_bf16_sq s GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES -- $PB
python tools/pmc_summary.py $O/pmc_bf16_fetch $O/pmc_bf16_write $O/pmc_bf16_sq $O/bf16_pmc_counters.json > /dev/null 2>>$O/pmc_summary.err
# instruction mix (segmenter workload): a pass with an unknown counter name just fails
PS="python bench.py --workload segmenter --steps 2 --warmup 1 --no-probe --no-cpu-baseline --no-sub --graph off"
pmc $O/pmc_seg_sq s GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -- $PS
pmc $O/pmc_seg_insts i SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES -- $PS
pmc $O/pmc_seg_insts2 j SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -- $PS
python tools/pmc_raw.py $O/pmc_seg_sq $O/pmc_seg_insts $O/pmc_seg_insts2 > $O/pmc_segmenter_raw.txt 2>&1; head -30 $O/pmc_segmenter_raw.txt
# the raw per-dispatch CSVs are tens of MB per pass: gpurun merges at most 64 MiB back (a PART B of round 3 lost everything to that limit)
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_bf16_fetch $O/pmc_bf16_write $O/pmc_bf16_sq $O/pmc_seg_sq $O/pmc_seg_insts $O/pmc_seg_insts2
du -sh $O
ls $O
fi
