#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_cli.py -x -q -k "several_ranks" > gpurun_out/c2_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c2_tests.log
timeout 900 python scratch/e2e_lab.py 16000000 2000 > gpurun_out/c2_lab.log 2>&1; echo "lab rc=$?"
cat gpurun_out/c2_lab.log | tail -80
