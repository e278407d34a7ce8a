#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c25_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c25_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scratch/r3_final.sh > gpurun_out/c25_final.log 2>&1; tail -45 gpurun_out/c25_final.log | cut -c1-200
