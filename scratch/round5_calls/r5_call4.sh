#!/bin/bash
# round 5, call 4: k_resolve_big on hardware: goldens (stress_pe has tuples with large smallest sets), then the stress workload with all pairs through the reference
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r5c4_pytest.log 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/r5c4_pytest.log
ARGS="--workload stress --pairs 4000000 --steps 3 --warmup 1 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --no-stress-leg --bootstraps 0"
timeout 900 python bench.py $ARGS --full-parity on --parity-sample 0 > gpurun_out/r5c4_stress.json 2> gpurun_out/r5c4_stress.log
echo "stress rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c4_stress.json'))
print(d['value'], d['ms_per_step'], d['breakdown_ms'])
print({k:v for k,v in d['parity_check_full_size'].items() if k not in ('reference','tolerance')})
PY
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stress -o t -- python $GRAFT_REPO_ROOT/bench.py $ARGS --full-parity off --parity-sample 0 --no-cpu-baseline > /tmp/prof_stress.json 2> /tmp/prof_stress.log
cd "$GRAFT_REPO_ROOT"
S=$(find /tmp/prof_stress -name '*kernel_stats.csv' | head -1)
python - "$S" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if not any(x in r['Name'] for x in ('at::','rocprim','hipcub'))]
with open('gpurun_out/r5c4_stress_kernel_stats.csv','w',newline='') as fo:
    w=csv.writer(fo); w.writerow(['Name','Calls','TotalDurationNs','AverageNs','MinNs','MaxNs'])
    for r in rows: w.writerow([r['Name'][:110],r['Calls'],r['TotalDurationNs'],r['AverageNs'],r['MinNs'],r['MaxNs']])
for r in rows[:16]: print(f"{r['Name'][:80]:80s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} ms  avg {float(r['AverageNs'])/1e3:10.2f} us")
PY
