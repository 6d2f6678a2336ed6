# round 4, GPU call: fp32 filter gradient with LDS-DMA staging (conv_wgrad_dma.hip) — parity, per-layer numbers, A/B on the joint step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -q -x 2>&1 | tail -3
for v in 1 0; do PNP_WGRAD_DMA=$v timeout 300 python tools/bench_conv.py > $O/conv_layers_dma$v.txt 2>/dev/null; done
paste <(cut -c1-28,66-84 $O/conv_layers_dma1.txt) <(cut -c66-84 $O/conv_layers_dma0.txt) | head -32
for v in 1 0 1 0; do
  PNP_WGRAD_DMA=$v timeout 600 python bench.py --no-cpu-baseline --no-probe --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('WGRAD_DMA=$v joint', r['value'], r['ms_per_step'], 'seg', r['segmenter_step']['value'])"
done
