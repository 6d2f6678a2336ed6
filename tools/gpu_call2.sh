cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_teacher_forced.py tests/test_gpu_dp.py tests/test_gpu_entrypoints.py tests/test_gpu_conv.py -m gpu -q --durations=8 > $O/tests.log 2>&1; tail -25 $O/tests.log
python tools/bench_conv.py > $O/conv_f32.txt 2>&1
PNP_LIB=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd/libpnp_hip_ilv.so python tools/bench_conv.py > $O/conv_f32_ilv.txt 2>&1
DTYPE=bf16 python tools/bench_conv.py > $O/conv_bf16.txt 2>&1
paste -d'\n' $O/conv_f32.txt $O/conv_f32_ilv.txt | head -60
cat $O/conv_bf16.txt
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; r=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('probe', r['value'], r['ms_per_step'], r['segmenter_step']['value'], r['roofline']['kernel'], r['roofline']['frac'])"
python bench.py --no-cpu-baseline --no-probe --no-sub 2> /dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('noprobe', r['value'], r['ms_per_step'])"
PNP_LIB=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd/libpnp_hip_ilv.so python bench.py --no-cpu-baseline --no-probe 2> /dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ilv noprobe', r['value'], r['ms_per_step'], r['segmenter_step']['value'])"
python bench.py --no-cpu-baseline --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err; python -c "
import json; r=json.loads(open('$O/bench_bf16.json').read().strip().splitlines()[-1]); print('bf16', r['value'], r['ms_per_step'], r['segmenter_step']['value'], r['roofline']['kernel'], r['roofline']['frac'])"; tail -2 $O/bench_bf16.err
