#!/bin/bash
# round 5: the whole -m gpu suite SERIALLY (VERDICT r4 #1b), log under gpurun_out/$R/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=${R:-r5s}; O=gpurun_out/$R; mkdir -p $O
timeout 1150 python -m pytest tests -m gpu -q -s --durations=15 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
grep -E "passed|failed|FAILED|slowest" $O/pytest_gpu.log | tail -12
