# kernel stats of 30 M stress pairs on the final build (3 steps + the small parity runs) -> gpurun_out/r06/r06_stress_full_size_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; rm -rf $O/strace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/strace -o t -- python $R/bench.py --workload stress --pairs 30000000 --steps 2 --warmup 1 --full-parity off --no-cpu-baseline --bootstraps 0 --parity-sample 200000 --detail-file /tmp/gb.json > /dev/null 2>&1
S=$(find $O/strace -name '*kernel_stats.csv' | head -1)
python - "$S" <<PY
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if not any(x in r['Name'] for x in ('at::','rocprim','hipcub'))]
with open('$O/r06_stress_full_size_kernel_stats.csv','w',newline='') as fo:
    w=csv.writer(fo); w.writerow(['Name','Calls','TotalDurationNs','AverageNs','MinNs','MaxNs'])
    for r in rows: w.writerow([r['Name'][:140],r['Calls'],r['TotalDurationNs'],r['AverageNs'],r['MinNs'],r['MaxNs']])
PY
rm -rf $O/strace
head -24 $O/r06_stress_full_size_kernel_stats.csv | cut -c1-75,140-230
