#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r02c
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -15
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02c/bench2.json 2> gpurun_out/r02c/bench2.err
python - <<'PY'
import json; d=json.load(open('/root/repo/gpurun_out/r02c/bench2.json')); print(d['value'], d['ms_per_step'], d['breakdown_ms'])
PY
