#!/bin/bash
cd /root/repo; O=gpurun_out/r02c; mkdir -p $O
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --parity-sample 0 > $O/v.json 2> $O/v.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r02c/v.json')); print(d['value'], d['ms_per_step'], d['breakdown_ms'], d['counters']['bucket_reads_per_pair'], d['roofline']['random_line_ceiling'])
PY
grep "index flattened\|on the device" $O/v.err | head -3
