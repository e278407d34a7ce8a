#!/bin/bash
cd /root/repo; O=gpurun_out/r02c; mkdir -p $O
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --parity-sample 0 --in-flight 2 > $O/v.json 2> $O/v.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r02c/v.json')); print(d['value'], d['ms_per_step'], d['breakdown_ms']); print(d.get('two_samples_in_flight'))
PY
tail -3 $O/v.err | cut -c1-300
