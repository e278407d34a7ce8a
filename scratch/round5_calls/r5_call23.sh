#!/bin/bash
# round 5, call 23: what is left of the GPU budget on a regression pass of the final build (all of test_gpu_parity.py, then the CLI tests as far as the time goes)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 40 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r5c23_parity.log 2>&1
echo "parity rc $?"; tail -2 gpurun_out/r5c23_parity.log
timeout 20 python -m pytest tests/test_gpu_cli.py -x -q > gpurun_out/r5c23_cli.log 2>&1
echo "cli rc $?"; tail -2 gpurun_out/r5c23_cli.log
