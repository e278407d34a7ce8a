#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_cli.py -x -q -k "human_pe" > gpurun_out/c30_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c30_tests.log
