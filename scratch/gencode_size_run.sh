#!/bin/bash
cd /root/repo; O=gpurun_out/lab; mkdir -p $O
timeout 900 python bench.py --genes 46000 --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 0 > $O/big.json 2> $O/big.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/lab/big.json')); print(d['value'], d['ms_per_step'], d['breakdown_ms'], d['counters'], d['roofline']['random_line_ceiling'], d['config']['workload'][:120])
PY
grep "index flattened\|transcriptome\|kallisto index" $O/big.err | head -4; tail -3 $O/big.err
