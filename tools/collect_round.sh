#!/bin/bash
# Regenerates the evidence under profiles/ for the current round on a GPU box:  R=r6z bash tools/collect_round.sh   (raw output gpurun_out/$R/; copy
# the summaries to profiles/rNN_*).  Round 6's version (round 5's is in the history at a60837a).
# PART=A: smoke, the driver's line (with cpu_baseline), segmenter / bf16 lines, kernel traces, per-layer tables, PMC passes (library symbols only:
# the joint step segfaults rocprofv3's counter collection otherwise), 8-rank same-device rehearsal.  PART=B: the whole -m gpu suite serially.  PART=C: the driver's line with the full cpu_baseline sample (2 warm-ups + median of 5).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=${R:-r6z}; O=gpurun_out/$R; mkdir -p $O
PART=${PART:-AB}
if [[ $PART == *A* ]]; then
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 600 $O/bench_n1.json; cp gpurun_out/bench_kernels_joint_f32.json $O/ 2>/dev/null
timeout 300 python bench.py --workload segmenter --no-sub --no-cpu-baseline > $O/bench_segmenter_n1.json 2>/dev/null; cp gpurun_out/bench_kernels_segmenter_f32.json $O/ 2>/dev/null
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16_n1.json 2>/dev/null
timeout 300 python bench.py --dtype bf16 --batch 32 --no-cpu-baseline --no-sub > $O/bench_bf16_B32_n1.json 2>/dev/null
# recorded steps at B = 16 (round 6: the bf16 one used to fault on replay) — slower than eager there, which is why --graph auto records at B <= 4 only
for dt in bf16 f32; do timeout 300 python bench.py --dtype $dt --graph on --no-cpu-baseline --no-sub --no-probe 2>/dev/null | tail -1; done > $O/bench_graph_on_B16_n1.json
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub --graph off"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_joint -o joint -- $B > $O/bench_prof_joint.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_seg -o seg -- $B --workload segmenter > $O/bench_prof_seg.json 2>/dev/null
for w in joint seg; do
  X=$(python -c "import json;r=json.loads(open('$O/bench_prof_$w.json').read().strip().splitlines()[-1]);print(r['steps']*r['ms_per_step'])" 2>/dev/null)
  python tools/rocpd_summary.py $(find $O/prof_$w -name "*.db" | head -1) $O/${w}_kernel_stats.txt --last-ms $X > /dev/null 2>&1
done
head -14 $O/joint_kernel_stats.txt | cut -c1-170
rm -rf $O/prof_joint $O/prof_seg
timeout 300 python tools/bench_conv.py > $O/conv_layers_f32.txt 2>/dev/null
PROF=1 ONLY="g5/6,g7,g8,g10,cls2 128,cls3 256,cls5" timeout 200 python tools/bench_conv.py 2>/dev/null > $O/per_kernel_layers.txt
# the route's GEMMs on the fp32 matrix pipe (X3=0) and on split-bf16 operands (X3=1: planner; X3=2: wherever the shapes allow), per kernel
for x3 in 0 1 2; do echo "== X3=$x3 (pnp_conv2d_wino_x3)"; X3=$x3 PROF=1 ONLY="g4 128,g5,g7,g8,g10,cls1 64,cls2 64,cls2 128,cls3 128,cls3 256,cls5" WINO=1 WINO_WGRAD=1 TILE=4 timeout 300 python tools/bench_conv.py 2>/dev/null; done > $O/x3_layers_B16.txt
# the narrow layers with the direct split-bf16 route off / on (conv_x3_direct.hip)
for v in 0 1; do echo "== PNP_X3_DIRECT=$v"; PNP_X3_DIRECT=$v PROF=1 ONLY="g2 32,g3,cls1 32,cls1 64,cls2 64" timeout 300 python tools/bench_conv.py 2>/dev/null; done > $O/x3_direct_layers_B16.txt
timeout 400 python -m pytest tests/test_gpu_trajectory.py -q -m gpu -s -p no:cacheprovider 2>&1 | grep -vE "amdgpu.ids|^make" > $O/trajectory.log
pmc() { local d=$1 o=$2 s=$3 rx=$4; shift 4; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -k 10 $s rocprofv3 --pmc "${ctr[@]}" --kernel-trace --kernel-include-regex "$rx" --output-format csv -d $d -o $o -- "$@" > $d.log 2>&1; echo "PMC pass $d rc=$?"; }
SQ="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES"
P1="python bench.py --steps 2 --warmup 1 --no-probe --no-cpu-baseline --no-sub --graph off"
RX="wino_|conv_|bn_|colreduce|splitk"
pmc $O/pmc_fetch f 200 "$RX" FETCH_SIZE -- $P1
pmc $O/pmc_write w 200 "$RX" WRITE_SIZE -- $P1
pmc $O/pmc_sq s 200 "$RX" $SQ -- $P1
PMC_NOTE="collected with --kernel-include-regex '$RX' (the joint step segfaults rocprofv3's counter collection when every dispatch is profiled)" python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_counters.json > /dev/null 2>$O/pmc_summary.err; tail -2 $O/pmc_summary.err
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq
for sc in weak strong; do
  bt=2; [ $sc = strong ] && bt=16
  PNP_DIST_BACKEND=gloo PNP_SAME_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --batch $bt --steps 2 --warmup 1 --no-cpu-baseline --scaling $sc > $O/dp8_same_device_$sc.json 2> $O/dp8_same_device_$sc.err; echo "dp8 $sc rc=$?"; tail -c 300 $O/dp8_same_device_$sc.json
done
# per-GPU batch sweep (strong-scaling operating points): eager / captured wall time against the sum of kernel durations, launches per step
R=$R bash tools/batch_sweep.sh f32 > $O/batch_sweep.log 2>&1; tail -8 $O/batch_sweep.log
fi
if [[ $PART == *B* ]]; then
timeout 1150 python -m pytest tests -m gpu -q -s --durations=15 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
grep -E "passed|failed|FAILED" $O/pytest_gpu.log | tail -6
fi
if [[ $PART == *C* ]]; then
# SURVEY 8(d)'s cpu_baseline sample in full: 2 warm-ups + the median of 5 timed oracle steps at B = 16 (about 9 minutes of host time)
timeout 1500 python bench.py --cpu-steps 5 --cpu-warmup 2 --cpu-small-batch 0 --no-sub > $O/bench_cpu_median5_n1.json 2> $O/bench_cpu_median5_n1.err; tail -c 500 $O/bench_cpu_median5_n1.json
fi
du -sh $O
