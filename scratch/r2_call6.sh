#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/c6
echo "=== GPU tests"
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
echo "=== kernel A"
PAIRS=30000000 VARIANTS="v3;v3:text_verify=2;v2" REPS=3 timeout 600 python scratch/ka_bench.py 2>&1 | grep -v amdgpu.ids | tail -5
echo "=== bench"
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/c6/bench.json 2> gpurun_out/c6/bench.log; tail -2 gpurun_out/c6/bench.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/c6/bench.json'))
print(d['value'], d['ms_per_step'], d['breakdown_ms'], d.get('parity_check',{}).get('ok'))
PY
