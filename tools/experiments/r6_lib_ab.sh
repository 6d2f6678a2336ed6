#!/bin/bash
# round 6: within-run A/B of two builds of the library (PNP_LIB): r6_lib_ab.sh <other .so> [bench flags]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OTHER=$PWD/$1; shift
for rep in 1 2; do for lib in "$OTHER" ""; do
  echo "== PNP_LIB=${lib:-<tree>} (rep $rep)"
  env ${lib:+PNP_LIB=$lib} timeout 600 python bench.py --no-cpu-baseline --no-sub --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms')"
done; done
