#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "long_reads or degenerate" 2>&1 | tail -12
