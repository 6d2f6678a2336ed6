cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
timeout 1200 python -m pytest tests -m "gpu and not slow" -q --durations=8 -x > $O/tests.log 2>&1; tail -15 $O/tests.log
DTYPE=bf16 python tools/bench_conv.py > $O/conv_bf16.txt 2>&1; cat $O/conv_bf16.txt
ONLY=512 python tools/bench_conv.py > $O/conv_f32.txt 2>&1
ONLY=512 PNP_LIB=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd/libpnp_hip_ilv.so python tools/bench_conv.py > $O/conv_f32_ilv.txt 2>&1
paste -d'\n' $O/conv_f32.txt $O/conv_f32_ilv.txt | head -30
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; r=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('f32', r['value'], r['ms_per_step'], r['segmenter_step']['value'])"
python bench.py --no-cpu-baseline --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err; python -c "
import json; r=json.loads(open('$O/bench_bf16.json').read().strip().splitlines()[-1]); print('bf16', r['value'], r['ms_per_step'], r['segmenter_step']['value'], r['roofline']['kernel'], r['roofline']['frac'], r['roofline_all_mfma_convs'])"; tail -2 $O/bench_bf16.err
