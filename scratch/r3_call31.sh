#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_gpu_cli.py tests/test_gpu_func_tests.py tests/test_gpu_fastq_units.py -x -q > gpurun_out/c31_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/c31_tests.log
