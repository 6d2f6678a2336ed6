# round 4, GPU call 1: hardware facts (LDS-DMA out-of-range lanes, ds_read_b64_tr_b16 lane map), first numbers of the bf16-resident
# forward / data-gradient kernels (all tiles), and one short bench.py run to see the compact last line as the driver will
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
timeout 60 tools/experiments/bin/micro_lds > $O/micro_lds.txt 2>&1; grep RESULT $O/micro_lds.txt
timeout 400 python tools/bench_bf16r.py > $O/bf16r_default.txt 2>&1; tail -32 $O/bf16r_default.txt
L="g7/9 512->512,g10 512,g5/6 256,cls2 128->128,cls3 256,g4 128->128"
for T in 0 1 2; do
  ONLY="$L" CHECK=0 PNP_BF16R_TILE=$T timeout 200 python tools/bench_bf16r.py > $O/bf16r_tile$T.txt 2>&1; echo "tile $T"; cat $O/bf16r_tile$T.txt | cut -c1-130
done
ONLY="g7/9 512->512,g10 512" CHECK=0 B=32 timeout 200 python tools/bench_bf16r.py > $O/bf16r_B32.txt 2>&1; cat $O/bf16r_B32.txt | cut -c1-130
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-steps 1 --cpu-warmup 0 --cpu-small-batch 0 > $O/bench_short.json 2> $O/bench_short.err
echo "bench rc=$? last line length: $(tail -1 $O/bench_short.json | wc -c)"; tail -1 $O/bench_short.json
