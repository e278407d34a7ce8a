#!/bin/bash
# the round's last call: the whole GPU suite, then the evidence for profiles/ (scratch/r4_final.sh)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r04/gpu_tests.log
bash scratch/r4_final.sh
