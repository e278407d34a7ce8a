#!/bin/bash
cd /root/repo; O=gpurun_out/r02c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -4
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --parity-sample 0 > $O/v.json 2> $O/v.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r02c/v.json')); print(d['value'], d['ms_per_step'], d['breakdown_ms'])
PY
