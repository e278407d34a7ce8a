#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r02c
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --parity-sample 0 > gpurun_out/r02c/bench3.json 2> gpurun_out/r02c/bench3.err
python - <<'PY'
import json; d=json.load(open('/root/repo/gpurun_out/r02c/bench3.json')); print(d['value'], d['ms_per_step'], d['breakdown_ms'])
PY
bash scratch/r2_call10.sh > /dev/null 2>&1
grep "k_tuple_slots\|k_bound_tuples\|k_resolve\|k_rec_dedup\|k_cand_singles\|step span" gpurun_out/r02b/timeline.txt | cut -c1-100
