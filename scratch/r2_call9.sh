#!/bin/bash
# f4: bus -x bulk / quant-tcc against the reference's goldens, plus the CLI tests that share the refactored front-end
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_bus_tcc.py tests/test_gpu_cli.py tests/test_gpu_func_tests.py -q 2>&1 | tail -40 > gpurun_out/call9_tests.log
cat gpurun_out/call9_tests.log
