#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r02b
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r02b/tests.log
cat gpurun_out/r02b/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err
cut -c1-700 gpurun_out/r02b/bench.json
KAMD_DEDUP_FORM=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02b/bench_dedup1.json 2> gpurun_out/r02b/bench_dedup1.err
cut -c1-400 gpurun_out/r02b/bench_dedup1.json
bash scratch/r2_call10.sh > /dev/null 2>&1
grep -v "copyBuffer\|fillBuffer\|k_em_sell" gpurun_out/r02b/timeline.txt | cut -c1-110 | head -70
