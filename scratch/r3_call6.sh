#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c6_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/c6_tests.log
timeout 600 python scratch/e2e_lab.py 16000000 2000 > gpurun_out/c6_lab.log 2>&1; echo "lab rc=$?"
grep -E "^==|wrote|written|gzip" gpurun_out/c6_lab.log | cut -c1-200
