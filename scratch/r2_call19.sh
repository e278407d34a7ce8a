#!/bin/bash
cd /root/repo; O=gpurun_out/r02c; mkdir -p $O
for gd in 4 16 64; do
KAMD_EM_GROUP_DIV=$gd timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --parity-sample 0 > $O/v.json 2> $O/v.err
grep "dbg" $O/v.err | head -3
done
