#!/bin/bash
# round 6: whole joint / segmenter step with one environment switch flipped (within-run A/B): r6_step.sh VAR "v1 v2 ..." [bench flags]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
VAR=$1; VALS=$2; shift 2
O=gpurun_out/r6step; mkdir -p $O
for rep in 1 2; do for v in $VALS; do
  echo "== $VAR=$v (rep $rep)" | tee -a $O/ab_$VAR.txt
  env $VAR=$v timeout 600 python bench.py --no-cpu-baseline --no-sub --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms', (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'))" | tee -a $O/ab_$VAR.txt
done; done
