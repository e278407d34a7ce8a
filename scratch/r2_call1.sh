#!/bin/bash
# round 2, GPU call 1: bring-up of the component-local EM (both kernels), the reference parity gate, a bench line
cd /root/repo
mkdir -p gpurun_out/c1
export KAMD_TEST_EXPERIMENTAL=1
for k in 1 2; do
  echo "=== local EM tests, KAMD_EML_KERNEL=$k"
  KAMD_EML_KERNEL=$k timeout 300 python -m pytest tests/test_gpu_parity.py -k "local" -x -q 2>&1 | tail -15
done
echo "=== EM forms on the bench workload"
PAIRS=30000000 timeout 600 python scratch/next_round/em_local_real.py 2>&1 | tail -12
echo "=== reference parity at the human-sized index"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "reference or oracle" 2>&1 | tail -8
echo "=== bench (streamed EM)"
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/c1/bench_streamed.json 2> gpurun_out/c1/bench_streamed.log; tail -3 gpurun_out/c1/bench_streamed.log; cat gpurun_out/c1/bench_streamed.json | cut -c1-1500
echo "=== bench (local EM, device plan)"
KAMD_EM_LOCAL=2 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/c1/bench_local.json 2> gpurun_out/c1/bench_local.log; tail -3 gpurun_out/c1/bench_local.log; cat gpurun_out/c1/bench_local.json | cut -c1-1200
