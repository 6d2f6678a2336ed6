#!/bin/bash
# round 4, Winograd route on by default: the three PMC passes of the joint step (FETCH_SIZE / WRITE_SIZE / SQ counters, each on its own)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4w4; mkdir -p $O
PB="python bench.py --steps 2 --warmup 1 --no-probe --no-cpu-baseline --no-sub --graph off"
PMC_OK=1
pmc() {
  [ "$PMC_OK" = 1 ] || return 0
  local d=$1 o=$2; shift 2; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -k 5 40 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $d -o $o -- "$@" > /dev/null 2>&1
  local rc=$?; if [ $rc -ge 124 ]; then echo "PMC pass $d timed out (rc $rc): skipping the remaining passes"; PMC_OK=0; fi
}
pmc $O/pmc_fetch f FETCH_SIZE -- $PB
pmc $O/pmc_write w WRITE_SIZE -- $PB
pmc $O/pmc_sq s GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES -- $PB
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_counters.json > /dev/null 2>$O/pmc_summary.err; tail -2 $O/pmc_summary.err
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4w4/pmc_counters.json'))
for k,v in d['kernels'].items():
    if 'wino' in k: print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
PY
