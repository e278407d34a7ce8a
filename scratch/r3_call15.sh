#!/bin/bash
# what bounds k_em_sell: timing experiments with a fixed 1280 rounds (results are garbage for EXP != 0)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --end-to-end 0 --no-pinned-pipeline --parity-sample 0"
for e in 0 1 2 3 6; do
  KAMD_EM_EXP=$e timeout 300 $B > gpurun_out/c16_e$e.json 2> gpurun_out/c16_e$e.err
  python - "$e" <<'P'
import json,sys
e=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/c16_e{e}.json').read().strip().splitlines()[-1]); print("exp", e, "em ms", d['breakdown_ms']['em'], "rounds", d['breakdown_ms']['em_rounds'], "step", d['ms_per_step'])
except Exception as ex: print(e, 'failed', ex); print(open(f'gpurun_out/c16_e{e}.err').read()[-600:])
P
done
