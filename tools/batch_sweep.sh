# Per-GPU-batch sweep of the joint step on ONE GPU (the strong-scaling operating points of SURVEY.md §8e: global B = 16 over 8 / 4 / 2 / 1
# GPUs = 2 / 4 / 8 / 16 slices per GPU): wall time per step with the step issued eagerly (~1 400 launches from Python) and as captured
# hipGraphs (step_capture.py), against the SUM of kernel durations of the same step (rocprofv3 --kernel-trace, timed region only).
#   bash tools/batch_sweep.sh [f32|bf16]  ->  gpurun_out/$R/batch_sweep_<dtype>.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${R:-r4z}; O=gpurun_out/$R; mkdir -p $O
D=${1:-f32}
echo "[" > $O/batch_sweep_$D.json
SEP=""
for B in 2 4 8 16; do
  C="python bench.py --dtype $D --batch $B --steps 20 --warmup 3 --no-cpu-baseline --no-probe --no-sub"
  E=$(timeout 300 $C --graph off 2>/dev/null | tail -1 | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')
  G=$(timeout 300 $C --graph on 2>/dev/null | tail -1 | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_b$B -o t -- $C --graph off --steps 5 > $O/prof_b$B.json 2>/dev/null
  X=$(python -c "import json;r=json.loads(open('$O/prof_b$B.json').read().strip().splitlines()[-1]);print(r['steps']*r['ms_per_step'])")
  python tools/rocpd_summary.py $(find $O/prof_b$B -name "*.db" | head -1) $O/kernels_b$B.txt --last-ms $X > /dev/null 2>&1
  KS=$(head -2 $O/kernels_b$B.txt | tail -1 | python -c 'import sys; t=sys.stdin.read().split(); print(float(t[4])/5.0, int(t[7])//5)')
  echo "$SEP{\"dtype\": \"$D\", \"per_gpu_batch\": $B, \"ms_per_step_eager\": $E, \"ms_per_step_captured\": $G, \"kernel_ms_per_step\": ${KS% *}, \"launches_per_step\": ${KS#* }}" >> $O/batch_sweep_$D.json
  SEP=","
  rm -rf $O/prof_b$B
done
echo "]" >> $O/batch_sweep_$D.json
cat $O/batch_sweep_$D.json
