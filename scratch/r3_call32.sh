#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 60 python -m pytest tests/test_gpu_func_tests.py tests/test_gpu_cli.py -x -q -k "func or ref_test_pe" > gpurun_out/c32_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/c32_tests.log
