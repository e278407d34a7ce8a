#!/bin/bash
# kernel-trace stats + step timeline of config #3 (the trace part of r6_final.sh on its own)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; rm -rf $O/trace
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --parity-sample 0 --no-stress-leg --no-gencode-leg --full-parity off --bootstraps 0 --detail-file /tmp/d2.json > /dev/null 2>&1
S=$(find $O/trace -name '*kernel_stats.csv' | head -1)
python - "$S" <<PY
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if not any(x in r['Name'] for x in ('at::','rocprim','hipcub'))]
with open('$O/r06_kernel_stats.csv','w',newline='') as fo:
    w=csv.writer(fo); w.writerow(['Name','Calls','TotalDurationNs','AverageNs','MinNs','MaxNs'])
    for r in rows: w.writerow([r['Name'][:140],r['Calls'],r['TotalDurationNs'],r['AverageNs'],r['MinNs'],r['MaxNs']])
PY
K=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python $R/scratch/timeline.py $K > $O/r06_step_timeline.txt 2>&1
rm -rf $O/trace
head -12 $O/r06_kernel_stats.csv | cut -c1-70,140-230
cd $R; timeout 600 python bench.py --extras --no-stress-leg --no-gencode-leg --full-parity off --bootstraps 0 --detail-file $O/r06_bench_extras_detail.json > $O/r06_bench_extras_line.json 2> $O/extras.log; cut -c1-400 $O/r06_bench_extras_line.json
