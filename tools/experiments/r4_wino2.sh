#!/bin/bash
# round 4, Winograd route on by default: the whole -m gpu suite (4 pytest-xdist workers, one file per worker at a time), then the joint bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4w2; mkdir -p $O
export TMPDIR=/tmp
timeout 840 python -m pytest tests -m gpu -q -n 4 --dist loadfile -p no:cacheprovider > $O/pytest_gpu_n4.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu_n4.log
grep -E "passed|failed|FAILED|ERROR" $O/pytest_gpu_n4.log | tail -30
