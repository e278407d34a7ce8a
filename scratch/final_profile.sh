#!/bin/bash
# one gpurun call: PMC traffic passes -> traffic.json, kernel-trace stats, then the bench line (which reads traffic.json)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/final; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
F=$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_write -name '*counter_collection.csv' | head -1)
python $R/tools/pmc_traffic.py $F $W $R/profiles r01 > $O/traffic.log 2>&1
cp $R/profiles/traffic.json $R/profiles/r01_pmc_fetch_size_per_kernel.csv $R/profiles/r01_pmc_write_size_per_kernel.csv $O/
rm -rf $O/pmc_fetch $O/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_trace.json 2> /dev/null
S=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp $S $O/kernel_stats.csv; rm -rf $O/trace
cd $R && python bench.py > $O/bench.json 2> $O/bench.log
cat $O/traffic.log; head -8 $O/kernel_stats.csv; cat $O/bench.json | cut -c1-400
