#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r03
timeout 1200 python bench.py --steps 20 --warmup 5 --bootstraps 100 > gpurun_out/r03/r03_bench.json 2> gpurun_out/r03/r03_bench.log; echo "bench rc=$?"
python - <<'P'
import json
b = json.loads(open("gpurun_out/r03/r03_bench.json").read().strip().splitlines()[-1])
e = b.get("end_to_end") or {}
print(b["value"], b["ms_per_step"], (b.get("parity_check") or {}).get("ok"), (b.get("parity_check_tail") or {}).get("ok"), b["breakdown_ms"])
print({k: {kk: v.get(kk) for kk in ("input_to_ecs_M_per_s", "whole_run_M_per_s", "wall_s", "after_index_s") if kk in v} if isinstance(v, dict) else v for k, v in e.items() if k not in ("note", "host")})
P
