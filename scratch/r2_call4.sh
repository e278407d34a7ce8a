#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/c4
echo "=== EM form tests"
timeout 600 python -m pytest tests/test_gpu_parity.py -k "em_forms" -x -q 2>&1 | tail -4
echo "=== EM shapes on the bench workload"
PAIRS=30000000 VARIANTS="local;local:em_split_len=32;local:em_split_len=16;local:em_split_len=8;local:em_local_block=1024;local:em_local_block=1024,em_split_len=32;local:em_local_block=1024,em_split_len=16;local:em_local_block=1024,em_split_len=8;local:em_local_block=1024,em_split_len=16,em_group_div=2;local:em_local_block=1024,em_split_len=16,em_group_div=3" timeout 900 python scratch/next_round/em_local_real.py 2>&1 | grep -v amdgpu.ids | tail -12
echo "=== rocprof kernel trace of one bench run"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/c4/prof -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/c4/bench_prof.json 2> /root/repo/gpurun_out/c4/bench_prof.log
cd /root/repo
find gpurun_out/c4/prof -name "*kernel_trace.csv" -size +30M -delete
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/c4/prof/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'at::' in n or 'rocprim' in n or 'hipcub' in n: continue
    print(f"{n[:60]:60s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:10.2f}")
PY
