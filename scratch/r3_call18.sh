#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_gpu_bus_tcc.py -x -q > gpurun_out/c18_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c18_tests.log
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pinned-pipeline"
timeout 900 $B > gpurun_out/c18_bench.json 2> gpurun_out/c18_bench.err; echo "bench rc=$?"
for e in 0 1 5; do
  KAMD_EM_EXP=$e timeout 300 $B --end-to-end 0 --parity-sample 0 --steps 3 --warmup 1 > gpurun_out/c18_e$e.json 2> gpurun_out/c18_e$e.err
  python - "$e" <<'P'
import json,sys
e=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/c18_e{e}.json').read().strip().splitlines()[-1]); print("exp", e, "em ms", d['breakdown_ms']['em'], "rounds", d['breakdown_ms']['em_rounds'], "step", d['ms_per_step'])
except Exception as ex: print(e, 'failed', ex); print(open(f'gpurun_out/c18_e{e}.err').read()[-600:])
P
done
python - <<'P'
import json
b = json.loads(open("gpurun_out/c18_bench.json").read().strip().splitlines()[-1])
print(b["value"], b["ms_per_step"], json.dumps(b["breakdown_ms"]))
pc = b.get("parity_check") or {}
print("parity", pc.get("ok"), pc.get("est_counts_max_rel_err_tpm_ge_1e-3"), pc.get("em_rounds_gpu"), (b.get("parity_check_tail") or {}).get("ok"))
e = b.get("end_to_end") or {}
print("e2e", {k: (v.get("input_to_ecs_M_per_s") if isinstance(v, dict) else v) for k, v in e.items() if k not in ("note", "host")})
P
