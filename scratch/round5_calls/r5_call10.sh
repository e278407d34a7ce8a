#!/bin/bash
# round 5, call 10: k_tup_absorb4 (four records per thread): parity suite, then the headline steps with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r5c10_pytest.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r5c10_pytest.log
ARGS="--steps 10 --warmup 2 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --no-stress-leg --bootstraps 0 --full-parity off --parity-sample 0 --no-cpu-baseline"
for v in four one; do
  if [ $v = one ]; then export KAMD_ABSORB_ONE=1; else unset KAMD_ABSORB_ONE; fi
  timeout 900 python bench.py $ARGS > gpurun_out/r5c10_$v.json 2> gpurun_out/r5c10_$v.log
  python - $v <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r5c10_{sys.argv[1]}.json')); print(sys.argv[1], d['value'], d['ms_per_step'], d['breakdown_ms'])
PY
done
