#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/lab; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_trace.json 2> $O/bench_trace.err
K=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python $R/scratch/timeline.py $K > $O/timeline.txt 2>&1
tail -40 $O/timeline.txt
rm -rf $O/trace
