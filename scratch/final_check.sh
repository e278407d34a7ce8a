#!/bin/bash
# what the driver does at round end: GPU tests, smoke, the default bench
cd /root/repo; mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 > gpurun_out/final/tests.log; cat gpurun_out/final/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/final/bench.json'))
print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['breakdown_ms'])
print('parity ok:', d['parity_check']['ok'], '| cpu', d['cpu_baseline']['value'], '| roofline', d['roofline']['frac'], d['roofline']['random_line_ceiling']['frac'])
PY
