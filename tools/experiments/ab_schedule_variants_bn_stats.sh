cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
P=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd
timeout 900 python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_segmenter.py tests/test_gpu_golden.py tests/test_gpu_conv.py tests/test_gpu_bf16.py -m "gpu and not slow" -q > $O/tests.log 2>&1; tail -8 $O/tests.log; grep "epilogue stats" $O/tests.log | head
for i in 1 2; do
for v in v0 "" v3 v4; do
L=$P/libpnp_hip${v:+_$v}.so
ONLY=512 PNP_LIB=$L python tools/bench_conv.py 2>/dev/null | grep "512" | sed "s/^/${v:-v2} /" >> $O/ab.txt
done; done
sort -k2,3 -s $O/ab.txt
for v in v0 "" v3 v4 "" v3; do
L=$P/libpnp_hip${v:+_$v}.so
PNP_LIB=$L python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${v:-v2}', r['value'], r['ms_per_step'], r['segmenter_step']['value'])"
done
PNP_FUSE_BN_STATS=0 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('v2 no epilogue stats', r['value'], r['ms_per_step'], r['segmenter_step']['value'])"
