#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 40 python -m pytest tests/test_gpu_cli.py -x -q -k "bound_to_the_library" > gpurun_out/c33_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/c33_tests.log
