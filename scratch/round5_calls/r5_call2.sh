#!/bin/bash
# round 5, call 2: hub fix + kernel trace of the stress workload (where do 146 ms of finalize and 64 ms of EM go?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "em_forms" > gpurun_out/r5c2_pytest.log 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/r5c2_pytest.log
ARGS="--workload stress --pairs 4000000 --steps 3 --warmup 1 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --no-stress-leg --bootstraps 0 --full-parity off --parity-sample 0 --no-cpu-baseline"
# first run builds and caches the index (38 s of the reference's `kallisto index`)
timeout 900 python bench.py $ARGS > gpurun_out/r5c2_stress.json 2> gpurun_out/r5c2_stress.log
echo "stress rc $?"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_stress -o stress -- python $GRAFT_REPO_ROOT/bench.py $ARGS > /tmp/prof_stress.json 2> /tmp/prof_stress.log
echo "rocprof rc $?"
cd "$GRAFT_REPO_ROOT"
find /tmp/prof_stress -name "*kernel_stats*" -exec cp {} gpurun_out/r5c2_stress_kernel_stats.csv \;
head -40 gpurun_out/r5c2_stress_kernel_stats.csv
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c2_stress.json'))
print(d['value'], d['ms_per_step'], d['breakdown_ms'])
print({k:d['counters'][k] for k in ('probes_per_pair','overflow_share','overflow_kernel_ms','em_form','em_oversized','em_plan_ms','em_largest_component_nnz')})
PY
