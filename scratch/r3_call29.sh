#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_cli.py tests/test_gpu_func_tests.py tests/test_gpu_fastq_units.py tests/test_gpu_bus_tcc.py -x -q > gpurun_out/c29_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c29_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
