#!/bin/bash
# round 5, call 4: cache tests; per-kernel per-layer times (PROF) for both tiles; PMC of the joint step restricted to the library's symbols
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5d; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_wino.py -q -s -x -p no:cacheprovider -k "cache or joint_steps" 2>&1 | tail -30) > $O/pytest_cache.log; tail -4 $O/pytest_cache.log; grep -i "joint steps: filter" $O/pytest_cache.log
L="g5/6,g7,g8,g10,cls2 128,cls3 256,cls5"
for t in 4 2; do PROF=1 ONLY="$L" WINO=2 WINO_WGRAD=2 TILE=$t timeout 150 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids > $O/prof_layers_tile$t.txt; done
cat $O/prof_layers_tile4.txt
grep -A12 "g7/9\|g10" $O/prof_layers_tile2.txt | head -60
pmc() { local d=$1 o=$2 s=$3 rx=$4; shift 4; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -k 10 $s rocprofv3 --pmc "${ctr[@]}" --kernel-trace --kernel-include-regex "$rx" --output-format csv -d $d -o $o -- "$@" > $d.log 2>&1; echo "PMC pass $d rc=$?"; }
SQ="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES"
P1="python bench.py --steps 2 --warmup 1 --no-probe --no-cpu-baseline --no-sub --graph off"
RX="wino_|conv_|bn_|colreduce|splitk"
pmc $O/pmc_fetch f 200 "$RX" FETCH_SIZE -- $P1
if [ ! -f $O/pmc_fetch/f_counter_collection.csv ] && ! ls $O/pmc_fetch/*/*counter_collection.csv > /dev/null 2>&1; then RX="wino_"; echo "fallback regex $RX"; pmc $O/pmc_fetch f 200 "$RX" FETCH_SIZE -- $P1; fi
pmc $O/pmc_write w 200 "$RX" WRITE_SIZE -- $P1
pmc $O/pmc_sq s 200 "$RX" $SQ -- $P1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_counters.json > /dev/null 2>$O/pmc_summary.err; tail -2 $O/pmc_summary.err
echo "regex: $RX" > $O/pmc_regex.txt
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5d/pmc_counters.json'))
for k,v in sorted(d['kernels'].items(), key=lambda kv: -kv[1].get('avg_dur_us_profiled',0)*kv[1].get('launches',0))[:16]:
    print(k[:60], {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('launches','hbm_read_MB_per_launch_corrected_x2','hbm_write_MB_per_launch','avg_dur_us_profiled','mfma_pipe_util')})
PY
