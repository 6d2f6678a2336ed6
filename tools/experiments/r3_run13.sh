# round 3, GPU call 13: wave priority for tile set-up / epilogue (s_setprio 3 around the non-MFMA parts; PNP_CONV_PRIO=0/1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
for t in 0 1; do PNP_CONV_PRIO=$t PNP_LIB=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd/libpnp_hip_trace.so timeout 100 python tools/experiments/stage_trace.py narrow wide 2>&1 | grep -v "amdgpu\|first instruction" > $O/trace_prio_$t.txt; cat $O/trace_prio_$t.txt; done
for t in 0 1; do PNP_CONV_PRIO=$t timeout 200 python tools/bench_conv.py 2>&1 | grep -v amdgpu > $O/conv_layers_prio_$t.txt; done
paste -d'|' <(cut -c1-62 $O/conv_layers_prio_0.txt) <(cut -c30-62 $O/conv_layers_prio_1.txt)
for t in 0 1 0 1; do
  PNP_CONV_PRIO=$t timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio=$t', r['value'], r['ms_per_step'], r['segmenter_step']['value'])" >> $O/ab.txt
done
cat $O/ab.txt
