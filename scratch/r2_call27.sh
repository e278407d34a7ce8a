#!/bin/bash
cd /root/repo; O=gpurun_out/r02c; mkdir -p $O
for cfg in "20000 0" "20000 1" "46000 1"; do
  set -- $cfg
  KAMD_TABLE_CONTIG=$2 timeout 900 python bench.py --genes $1 --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 0 > $O/big.json 2> $O/big.err
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open('/root/repo/gpurun_out/r02c/big.json')); print(sys.argv[1], d['value'], d['ms_per_step'], d['breakdown_ms']['pseudoalign_kernel'], d['roofline']['random_line_ceiling'].get('GB/s_in_64B_lines'))
PY
  grep "kamd\]" $O/big.err | head -3
done
