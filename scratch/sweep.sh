#!/bin/bash
# usage: scratch/sweep.sh VAR v1 v2 ... ; prints kernel A ms per value
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], d['breakdown_ms'])"
done
