# round 4, GPU call 2: the bf16-resident filter gradient (ds_read_b64_tr_b16 fragments) next to forward / data gradient, every layer
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
timeout 600 python tools/bench_bf16r.py > $O/bf16r_all.txt 2>&1; tail -34 $O/bf16r_all.txt | cut -c1-220
