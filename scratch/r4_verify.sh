#!/bin/bash
# the round's closing check: the whole GPU suite and the default bench line on the final build
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04/gpu_tests.log | head -1
timeout 1200 python bench.py --steps 20 --warmup 5 --bootstraps 100 > gpurun_out/r04/r04_bench.json 2> gpurun_out/r04/r04_bench.log; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04/r04_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['breakdown_ms'], d['roofline']['frac'])
for k in ('parity_check', 'parity_check_tail', 'parity_check_full_size'): print(k, d[k]['ok'])
e = d['end_to_end']; print({k: (v.get('input_to_ecs_M_per_s'), v.get('wall_s'), v.get('index_load_s')) for k, v in e.items() if isinstance(v, dict) and 'wall_s' in v})
PY
