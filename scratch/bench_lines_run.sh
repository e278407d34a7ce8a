#!/bin/bash
# the two bench lines kept under profiles/ (configs #3 + #5 + end-to-end + two samples in flight; config #2)
cd /root/repo; O=gpurun_out/r02; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 --end-to-end 4000000 --bootstraps 100 --in-flight 2 > $O/r02_bench.json 2> $O/r02_bench.log
timeout 600 python bench.py --workload yeast --steps 10 --warmup 3 --end-to-end 4000000 > $O/r02_bench_config2_yeast.json 2> $O/r02_bench_config2_yeast.log
cut -c1-300 $O/r02_bench.json; echo; cut -c1-300 $O/r02_bench_config2_yeast.json
