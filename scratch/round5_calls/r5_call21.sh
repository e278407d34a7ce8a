#!/bin/bash
# round 5, call 21: plan step G by one range per (workgroup, group), k_comp_stats without the atomic queue: EM + multi-rank parity tests, then config #3's kernel table
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "em or reproducible or bootstrap or stress" > gpurun_out/r5c21_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r5c21_pytest.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5c21_prof -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --full-parity off --no-stress-leg --no-gencode-leg --no-config2 > gpurun_out/r5c21_bench.log 2>&1
echo "bench rc $?"; grep '^{' gpurun_out/r5c21_bench.log | cut -c1-300
for f in $(find gpurun_out/r5c21_prof -name '*kernel_stats.csv'); do
  if grep -q k_match_v3 "$f"; then cp "$f" gpurun_out/r5c21_kernel_stats.csv; fi
done
grep -E "k_eml_|k_cc_union|k_comp_stats|k_sell_" gpurun_out/r5c21_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/r5c21_prof
