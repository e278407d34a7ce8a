#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_fastq_units.py tests/test_gpu_cli.py tests/test_gpu_bus_tcc.py tests/test_gpu_multirank.py tests/test_gpu_func_tests.py -x -q > gpurun_out/c11_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/c11_tests.log
timeout 600 python scratch/e2e_lab.py 16000000 2000 > gpurun_out/c11_lab.log 2>&1; echo "lab rc=$?"
grep -E "^==|wrote|written|gzip|device parser" gpurun_out/c11_lab.log | cut -c1-330 | head -30
