# round-2 A/B: ring depth of the filter gradient, two-launch BN final combine, stride phases in one launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_elementwise.py tests/test_gpu_dp.py -m "gpu and not slow" -q -x > $O/tests.log 2>&1; tail -4 $O/tests.log
for d in 1 2 3; do PNP_WGRAD_DEPTH=$d python tools/bench_conv.py 2>/dev/null | sed "s/^/depth$d /" > $O/conv_depth$d.txt; done
paste -d'\n' $O/conv_depth1.txt $O/conv_depth2.txt $O/conv_depth3.txt | cut -c1-40,92-112
run() { # label, env...
  L=$1; shift
  env "$@" python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(r['value'],2), round(r['ms_per_step'],2), round(r['segmenter_step']['value'],1))" | tee -a $O/ab.txt
}
run base PNP_WGRAD_DEPTH=1 PNP_BN_FINAL_1STAGE=1 PNP_CONV_NOPHASEGROUP=1
run ring2 PNP_WGRAD_DEPTH=2 PNP_BN_FINAL_1STAGE=1 PNP_CONV_NOPHASEGROUP=1
run ring3 PNP_WGRAD_DEPTH=3 PNP_BN_FINAL_1STAGE=1 PNP_CONV_NOPHASEGROUP=1
run final2 PNP_WGRAD_DEPTH=1 PNP_CONV_NOPHASEGROUP=1
run group PNP_WGRAD_DEPTH=1 PNP_BN_FINAL_1STAGE=1
run all2 PNP_WGRAD_DEPTH=2
run all3 PNP_WGRAD_DEPTH=3
run base PNP_WGRAD_DEPTH=1 PNP_BN_FINAL_1STAGE=1 PNP_CONV_NOPHASEGROUP=1
run all2 PNP_WGRAD_DEPTH=2
