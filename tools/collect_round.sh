cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1f
if [ -z "$SKIP_TESTS" ]; then timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r1f/tests.log 2>&1; tail -2 gpurun_out/r1f/tests.log; fi
python bench.py > gpurun_out/r1f/bench.json 2> gpurun_out/r1f/bench.err; tail -c 600 gpurun_out/r1f/bench.json
rocprofv3 --kernel-trace --stats -d gpurun_out/r1f/prof -o seg -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r1f/bench_prof.json 2>/dev/null
B="python bench.py --steps 2 --warmup 1 --no-probe --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r1f/pmc_fetch -o f -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r1f/pmc_write -o w -- $B > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/r1f/pmc_sq -o s -- $B > /dev/null 2>&1
python tools/bench_conv.py > gpurun_out/r1f/conv_layers.txt 2>/dev/null
python tools/bench_gan.py 2>/dev/null | tail -1 > gpurun_out/r1f/gan.json; cat gpurun_out/r1f/gan.json
STEPS=3 rocprofv3 --kernel-trace --stats -d gpurun_out/r1f/prof_gan -o gan -- python tools/bench_gan.py > /dev/null 2>&1
python bench.py --workload gan --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r1f/bench_gan_workload.json
python tools/e2e_segmenter.py 2>&1 | grep "E2E" > gpurun_out/r1f/e2e.txt; cat gpurun_out/r1f/e2e.txt
ls gpurun_out/r1f
