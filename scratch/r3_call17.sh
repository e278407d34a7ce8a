#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -x -q > gpurun_out/c17_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c17_tests.log
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --end-to-end 0 --no-pinned-pipeline"
timeout 600 $B > gpurun_out/c17_bench.json 2> gpurun_out/c17_bench.err; echo "bench rc=$?"
for e in 0 1 5; do
  KAMD_EM_EXP=$e timeout 300 $B --parity-sample 0 --steps 3 --warmup 1 > gpurun_out/c17_e$e.json 2> gpurun_out/c17_e$e.err
  python - "$e" <<'P'
import json,sys
e=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/c17_e{e}.json').read().strip().splitlines()[-1]); print("exp", e, "em ms", d['breakdown_ms']['em'], "rounds", d['breakdown_ms']['em_rounds'], "step", d['ms_per_step'])
except Exception as ex: print(e, 'failed', ex); print(open(f'gpurun_out/c17_e{e}.err').read()[-600:])
P
done
python - <<'P'
import json
b = json.loads(open("gpurun_out/c17_bench.json").read().strip().splitlines()[-1])
print(b["value"], b["ms_per_step"], json.dumps(b["breakdown_ms"]))
pc = b.get("parity_check") or {}
print("parity", pc.get("ok"), pc.get("est_counts_max_rel_err_tpm_ge_1e-3"), pc.get("em_rounds_gpu"), (b.get("parity_check_tail") or {}).get("ok"))
P
