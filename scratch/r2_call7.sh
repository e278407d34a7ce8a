#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/c7
echo "=== GPU tests"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
echo "=== bench + bootstraps"
timeout 900 python bench.py --steps 5 --warmup 2 --bootstraps 100 > gpurun_out/c7/bench_b100.json 2> gpurun_out/c7/bench_b100.log; tail -2 gpurun_out/c7/bench_b100.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/c7/bench_b100.json'))
print(d['value'], d['ms_per_step'], d['breakdown_ms'], d.get('parity_check',{}).get('ok')); print(d.get('bootstrap'))
PY
