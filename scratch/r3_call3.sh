#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{ echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cfs_quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) period $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"; nproc; lscpu | grep -E "Model name|Socket|NUMA node\(s\)|Thread|Core"; grep -E "nr_throttled|throttled" /sys/fs/cgroup/cpu.stat /sys/fs/cgroup/cpu/cpu.stat 2>/dev/null; } > gpurun_out/c3_host.txt 2>&1
cat gpurun_out/c3_host.txt
timeout 600 python -m pytest tests/test_gpu_fastq_units.py tests/test_gpu_cli.py -x -q > gpurun_out/c3_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/c3_tests.log
timeout 900 python scratch/e2e_lab.py 16000000 2000 > gpurun_out/c3_lab.log 2>&1; echo "lab rc=$?"
grep -E "^==|device parser|wrote|written|gzip" gpurun_out/c3_lab.log | cut -c1-330
grep -E "throttled" /sys/fs/cgroup/cpu.stat /sys/fs/cgroup/cpu/cpu.stat 2>/dev/null
