#!/bin/bash
# round 5, call 19: the plan's steps B / F with one atomic per (wavefront, group): EM parity tests, then the kernel table of config #3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -k "em or reproducible or bootstrap" > gpurun_out/r5c19_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r5c19_pytest.log
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5c19_prof -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --full-parity off --no-stress-leg --no-gencode-leg --no-config2 > gpurun_out/r5c19_bench.log 2>&1
echo "bench rc $?"; tail -1 gpurun_out/r5c19_bench.log | cut -c1-400
f=$(find gpurun_out/r5c19_prof -name '*kernel_stats.csv' | head -1)
grep -E "k_eml_step|k_cc_union|k_eml_group_build|k_eml_rank" "$f" | cut -c1-160
cp "$f" gpurun_out/r5c19_kernel_stats.csv
rm -rf gpurun_out/r5c19_prof
