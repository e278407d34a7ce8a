#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r02c
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -30 > gpurun_out/r02c/tests.log
tail -30 gpurun_out/r02c/tests.log
