#!/bin/bash
# round 5, call 14: why do the hot tables not pay?  counts + kernel stats at 30 M stress pairs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--workload stress --steps 2 --warmup 1 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --no-stress-leg --no-gencode-leg --bootstraps 0 --full-parity off --parity-sample 0 --no-cpu-baseline"
timeout 900 python bench.py $ARGS > /dev/null 2>&1
cd /tmp && KAMD_DEBUG_FIN=1 timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stress -o t -- python $GRAFT_REPO_ROOT/bench.py $ARGS > /tmp/o.json 2> /tmp/prof_stress.log
cd "$GRAFT_REPO_ROOT"
grep "kamd\] hybrid" /tmp/prof_stress.log | tail -1
S=$(find /tmp/prof_stress -name '*kernel_stats.csv' | head -1)
python - "$S" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if not any(x in r['Name'] for x in ('at::','rocprim','hipcub'))]
for r in rows[:8]: print(f"{r['Name'][:80]:80s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} ms  avg {float(r['AverageNs'])/1e3:10.2f} us")
PY
