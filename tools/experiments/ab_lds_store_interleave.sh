cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
ILV=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd/libpnp_hip_ilv.so
for i in 1 2 3; do
ONLY=512 python tools/bench_conv.py 2>/dev/null | grep "512" | sed 's/^/A /' >> $O/ab.txt
ONLY=512 PNP_LIB=$ILV python tools/bench_conv.py 2>/dev/null | grep "512" | sed 's/^/B /' >> $O/ab.txt
done
sort -k2,3 -s $O/ab.txt
timeout 1200 python -m pytest tests -m "gpu and not slow" -q --durations=6 > $O/tests.log 2>&1; tail -12 $O/tests.log
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; r=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('f32', r['value'], r['ms_per_step'], r['segmenter_step']['value'])"
PNP_LIB=$ILV python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ilv2', r['value'], r['ms_per_step'], r['segmenter_step']['value'])"
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 again', r['value'], r['ms_per_step'], r['segmenter_step']['value'])"
PNP_LIB=$ILV python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ilv2 again', r['value'], r['ms_per_step'], r['segmenter_step']['value'])"
