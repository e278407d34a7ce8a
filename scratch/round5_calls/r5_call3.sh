#!/bin/bash
# round 5, call 3: kernel trace of the stress workload
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--workload stress --pairs 4000000 --steps 3 --warmup 1 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --no-stress-leg --bootstraps 0 --full-parity off --parity-sample 0 --no-cpu-baseline"
timeout 900 python bench.py $ARGS > /dev/null 2> /dev/null     # (builds and caches the index)
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stress -o t -- python $GRAFT_REPO_ROOT/bench.py $ARGS > /tmp/prof_stress.json 2> /tmp/prof_stress.log
echo "rocprof rc $?"
cd "$GRAFT_REPO_ROOT"
S=$(find /tmp/prof_stress -name '*kernel_stats.csv' | head -1)
python - "$S" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if not any(x in r['Name'] for x in ('at::','rocprim','hipcub'))]
with open('gpurun_out/r5c3_stress_kernel_stats.csv','w',newline='') as fo:
    w=csv.writer(fo); w.writerow(['Name','Calls','TotalDurationNs','AverageNs','MinNs','MaxNs'])
    for r in rows: w.writerow([r['Name'][:110],r['Calls'],r['TotalDurationNs'],r['AverageNs'],r['MinNs'],r['MaxNs']])
for r in rows[:45]: print(f"{r['Name'][:90]:90s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} ms  avg {float(r['AverageNs'])/1e3:10.2f} us")
PY
