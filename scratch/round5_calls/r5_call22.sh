#!/bin/bash
# round 5, call 22 (the last): the final build's EM parity tests and its headline on config #3 (20 steps; no child legs, no end-to-end legs, sample parity only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 python -m pytest tests/test_gpu_parity.py -x -q -k "em_forms or reproducible" > gpurun_out/r5c22_pytest.log 2>&1
echo "pytest rc $?"; tail -2 gpurun_out/r5c22_pytest.log
timeout 140 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --full-parity off --no-stress-leg --no-gencode-leg --no-config2 --no-compact-leg --end-to-end 0 --no-pinned-pipeline > gpurun_out/r5c22_bench.log 2> gpurun_out/r5c22_bench.err
echo "bench rc $?"; grep '^{' gpurun_out/r5c22_bench.log | cut -c1-400
