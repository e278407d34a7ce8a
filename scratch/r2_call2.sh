#!/bin/bash
# round 2, GPU call 2: kernel A v3 (unitig text, 24 waves/CU) against v2; the whole GPU test-suite; EM-local kernel trace
cd /root/repo
mkdir -p gpurun_out/c2
echo "=== kernel A variants"
PAIRS=30000000 VARIANTS="v2;v3;v3:text_verify=2;v3:refill_min=4;v3:refill_min=16;v3:items_per_wave=512;v3:items_per_wave=2048" timeout 900 python scratch/ka_bench.py 2>&1 | tail -12
echo "=== GPU tests"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "=== rocprof: EM local + kernel A v3"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/c2/prof -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/c2/bench_prof.json 2> /root/repo/gpurun_out/c2/bench_prof.log
cd /root/repo
f=$(find gpurun_out/c2/prof -name "*kernel_stats.csv" | head -1); echo $f; head -40 $f
cat gpurun_out/c2/bench_prof.json | cut -c1-900
find gpurun_out/c2/prof -name "*kernel_trace.csv" -size +30M -delete
