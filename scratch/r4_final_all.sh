#!/bin/bash
# the round's last call: the whole GPU suite, then the evidence for profiles/ (scratch/r4_final.sh)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r04/gpu_tests.log
bash scratch/r4_final.sh
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --parity-sample 0 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --bootstraps 0 --full-parity off --in-flight 2 > gpurun_out/r04/r04_bench_two_samples_in_flight.json 2> gpurun_out/r04/in_flight.err
python -c "import json; d=json.loads(open('gpurun_out/r04/r04_bench_two_samples_in_flight.json').read().strip().splitlines()[-1]); print('in flight:', d['value'], d.get('two_samples_in_flight'))"
