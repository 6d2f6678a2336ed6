cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2i; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_elementwise.py tests/test_gpu_segmenter.py tests/test_gpu_adversarial.py tests/test_gpu_dp.py tests/test_gpu_teacher_forced.py -m "gpu and not slow" -q -x > $O/tests.log 2>&1; tail -5 $O/tests.log
run() { L=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-probe 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(r['value'],2), round(r['ms_per_step'],2), round(r['segmenter_step']['value'],1))" | tee -a $O/ab.txt; }
run all X=1
run nosink PNP_GRAD_SINKS=0
run nolink PNP_RES_LINK=0
run neither PNP_GRAD_SINKS=0 PNP_RES_LINK=0
run all X=1
run neither PNP_GRAD_SINKS=0 PNP_RES_LINK=0
