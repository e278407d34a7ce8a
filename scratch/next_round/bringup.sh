#!/bin/bash
# first GPU call of the next round: the experimental EM form, small tests first (wrap in `timeout`: none of this has run on hardware)
cd /root/repo
KAMD_TEST_EXPERIMENTAL=1 timeout 120 python -m pytest tests/test_gpu_parity.py -k "local" -x -q 2>&1 | tail -15
PAIRS=8000000 timeout 200 python scratch/next_round/em_local_real.py 2>&1 | tail -8
