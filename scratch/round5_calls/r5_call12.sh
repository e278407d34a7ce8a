#!/bin/bash
# round 5, call 12: the whole GPU suite, then the evidence for profiles/ (PMC traffic, counters, kernel stats, timeline, the bench lines)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
SECONDS=0
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r5c12_pytest.log 2>&1
echo "pytest rc $? in $SECONDS s"; tail -4 gpurun_out/r5c12_pytest.log
SECONDS=0
bash scratch/r5_final.sh > gpurun_out/r5c12_final.log 2>&1
echo "final rc $? in $SECONDS s"; tail -60 gpurun_out/r5c12_final.log | cut -c1-300
