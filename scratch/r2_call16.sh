#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r02/tests.log
cat gpurun_out/r02/tests.log
if grep -q " passed" gpurun_out/r02/tests.log && ! grep -q "failed" gpurun_out/r02/tests.log; then bash scratch/r2_profile.sh; fi
