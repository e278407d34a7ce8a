#!/bin/bash
cd /root/repo; O=gpurun_out/r02c; mkdir -p $O
KAMD_DEBUG_TUPLES=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --parity-sample 0 > $O/v.json 2> $O/v.err
grep "dbg" $O/v.err | head -2
