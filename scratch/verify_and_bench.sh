#!/bin/bash
R=/root/repo; O=$R/gpurun_out/final2; mkdir -p $O
cd $R && python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?
tail -3 $O/tests.log
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $O/tests.log | head -20; exit 1; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_trace.json 2> /dev/null
S=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp $S $O/kernel_stats.csv; rm -rf $O/trace
cd $R && python bench.py > $O/bench.json 2> $O/bench.log
grep -n "k_match_v2\|k_pm_" $O/kernel_stats.csv | cut -c1-60,200-330 | head; cut -c1-700 $O/bench.json
