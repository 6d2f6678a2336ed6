# round 4, GPU call 4: bf16 parity tests (resident + staged), bf16 lines with the bf16-only accumulator gradient, A/B inside one call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bf16r.py tests/test_gpu_bf16.py -q -s > $O/tests_bf16.log 2>&1; tail -5 $O/tests_bf16.log
grep "resident vs\|joint step bf16\|bf16 vs fp32\|bf16 losses" $O/tests_bf16.log | cut -c1-300
for v in 1 0 1 0; do
  PNP_BF16_ONLY_H=$v timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --no-probe > $O/bench_bf16_onlyh$v.json 2>/dev/null
  echo "ONLY_H=$v $(tail -1 $O/bench_bf16_onlyh$v.json | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"], r["segmenter_step"]["value"])')"
done
