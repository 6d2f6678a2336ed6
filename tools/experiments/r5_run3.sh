#!/bin/bash
# round 5, call 3: transformed-filter cache + XCD mappings — parity, within-run A/B per layer and on the joint step, PMC retry variants
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c; mkdir -p $O
(timeout 400 python -m pytest tests/test_gpu_wino.py tests/test_gpu_capture.py -q -s -x -p no:cacheprovider 2>&1 | tail -40) > $O/pytest_wino_capture.log; tail -4 $O/pytest_wino_capture.log
grep -i "joint steps: filter" $O/pytest_wino_capture.log
L="g5,g7,g8,g10,cls2 128,cls3,cls5"
for cfg in "1 0" "2 0" "2 1" "1 1"; do set -- $cfg
  echo "== PNP_WINO_XCD=$1 PNP_WINO_XCD_IN=$2"
  PNP_WINO_XCD=$1 PNP_WINO_XCD_IN=$2 ONLY="$L" WINO=2 WINO_WGRAD=2 TILE=4 timeout 120 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_xcd$1_in$2.txt
done
for sp in 1 2; do echo "== PNP_WINO_WGRAD_SPLIT=$sp"; PNP_WINO_WGRAD_SPLIT=$sp ONLY="g7/9,g5/6,cls3 256" WINO=2 WINO_WGRAD=2 TILE=4 timeout 60 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids; done
run() {  # run <tag> <env...>
  local tag=$1; shift
  env "$@" timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "
import json; r=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag joint', r['value'], r['ms_per_step'], 'segmenter', r['segmenter_step']['value'], 'bf16', r.get('bf16_step',{}).get('value'), r['roofline']['kernel'], r['roofline']['frac'], len(json.dumps(r)))" || tail -5 $O/bench_$tag.err
}
run base PNP_WINO_XCD=1 PNP_WINO_XCD_IN=0 PNP_WINOGRAD_UCACHE=0
run new X=1
run base2 PNP_WINO_XCD=1 PNP_WINO_XCD_IN=0 PNP_WINOGRAD_UCACHE=0
run new2 X=1
run nocache PNP_WINOGRAD_UCACHE=0
cp gpurun_out/bench_kernels_joint_f32.json $O/ 2>/dev/null
# PMC on a whole step: which variant survives (the joint step segfaults rocprofv3's counter collection, r5b)
pmc() { local d=$1 o=$2 s=$3; shift 3; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -k 10 $s rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $d -o $o -- "$@" > $d.log 2>&1; echo "PMC pass $d rc=$?"; }
pmc $O/pmcS_fetch f 120 FETCH_SIZE -- python bench.py --workload segmenter --steps 2 --warmup 1 --no-probe --no-cpu-baseline --no-sub --graph off
ls $O/pmcS_fetch 2>/dev/null | head -3
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "wino" --output-format csv -d $O/pmcR_fetch -o f -- python bench.py --steps 2 --warmup 1 --no-probe --no-cpu-baseline --no-sub --graph off > $O/pmcR_fetch.log 2>&1; echo "PMC regex pass rc=$?"
find $O/pmcR_fetch -name "*counter_collection.csv" | head -2
du -sh $O/pmcS_fetch $O/pmcR_fetch 2>/dev/null
