#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/lab
export KAMD_BENCH_SHARE_GPU=1 KAMD_BENCH_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --pairs 3000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/lab/share2.json 2> gpurun_out/lab/share2.err
echo "exit code $?"
grep -c "bad_variant\|Traceback\|FAILED" gpurun_out/lab/share2.err
python - <<'PY'
import json
s=open('/root/repo/gpurun_out/lab/share2.json').read(); d=json.loads(s[s.index('{"metric"'):s.rindex('}')+1])
print(d['value'], d['n_gpus'], d['scaling'], d['ms_per_step'], d['breakdown_ms'], d.get('other_scaling',{}).get('value'))
PY
