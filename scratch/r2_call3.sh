#!/bin/bash
# round 2, GPU call 3: sliced-ELLPACK EM kernel (shape sweep), kernel A variants (text on/off, v2 fixed), the GPU test-suite
cd /root/repo
mkdir -p gpurun_out/c3
echo "=== EM form tests"
timeout 600 python -m pytest tests/test_gpu_parity.py -k "em_forms" -x -q 2>&1 | tail -8
echo "=== EM forms on the bench workload"
PAIRS=30000000 timeout 900 python scratch/next_round/em_local_real.py 2>&1 | grep -v amdgpu.ids | tail -14
echo "=== kernel A variants"
PAIRS=30000000 VARIANTS="v2;v3;v3:text_verify=2;v3:lds_pad=8192;v3:lds_pad=16384" timeout 900 python scratch/ka_bench.py 2>&1 | grep -v amdgpu.ids | tail -8
echo "=== GPU tests"
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
echo "=== bench"
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/c3/bench.json 2> gpurun_out/c3/bench.log; tail -2 gpurun_out/c3/bench.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/c3/bench.json'))
print(d['value'], d['ms_per_step'], d['breakdown_ms'], d.get('parity_check',{}).get('ok'))
PY
