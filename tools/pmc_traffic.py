#!/usr/bin/env python
"""Per-kernel HBM-side traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately as
MI355X_MICROARCH.md prescribes) of `python bench.py --steps 1 --warmup 0 --no-cpu-baseline`.

usage: tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out_dir> <prefix> [pairs [em_rounds_executed]]

Writes <out_dir>/<prefix>_pmc_fetch_size_per_kernel.csv, <out_dir>/<prefix>_pmc_write_size_per_kernel.csv and
<out_dir>/traffic.json (read by bench.py into roofline.traffic):
    kernel_a_hbm_bytes = (FETCH_SIZE + WRITE_SIZE) x 1024 B of the one k_match_v3 launch
    em_round_hbm_bytes = the same for the EM pass kernels, summed, divided by the number of rounds
Both counters are in KB.  Raw values are used (no x2): see profiles/README.md for the calibration against the kernel's
own count of 64-byte bucket reads."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def per_kernel(path, counter):
    tot, calls = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        calls[k] += 1
    return tot, calls


def main():
    fetch, write, out_dir, prefix = sys.argv[1:5]
    pairs = int(sys.argv[5]) if len(sys.argv) > 5 else 30_000_000
    res = {}
    for path, counter, tag in ((fetch, "FETCH_SIZE", "fetch"), (write, "WRITE_SIZE", "write")):
        tot, calls = per_kernel(path, counter)
        res[tag] = (tot, calls)
        with open(f"{out_dir}/{prefix}_pmc_{tag}_size_per_kernel.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", f"{counter}_total_KB", f"{counter}_per_call_KB"])
            for k in sorted(tot, key=lambda k: -tot[k])[:40]:
                w.writerow([k, calls[k], round(tot[k], 1), round(tot[k] / calls[k], 3)])

    def kb(pred):
        return sum(res[t][0][k] for t in ("fetch", "write") for k in res[t][0] if pred(k))

    a_kb = kb(lambda k: k.startswith("k_match_v3") or k.startswith("k_match_v2") or k.startswith("k_pseudoalign<"))
    em_pred = lambda k: k.startswith("k_pm_rows_pass") or k.startswith("k_pm_cols_pass") or k.startswith("k_pm_rows_fix") or \
        k.startswith("k_pm_cols_fix") or k.startswith("k_em_rows") or k.startswith("k_em_seg") or k.startswith("k_em_final") or \
        k.startswith("k_em_sell") or k.startswith("k_em_local")
    em_kb = kb(em_pred)
    fin_pred = lambda k: k.startswith("k_tup_") or k.startswith("k_rec_dedup") or k.startswith("k_rec_insert") or k.startswith("k_rec_verify") or k.startswith("k_bound_") or k.startswith("k_resolve") or \
        k.startswith("k_cand_singles") or k.startswith("k_final_") or k.startswith("k_table_init") or k.startswith("k_scan_")
    fin_kb = kb(fin_pred)
    calls = res["fetch"][1]
    local_launches = sum(calls[k] for k in calls if k.startswith("k_em_sell") or k.startswith("k_em_local"))
    rounds = max([calls[k] for k in calls if k.startswith("k_pm_rows_pass") or k.startswith("k_em_rows")] + [1])
    if local_launches:   # component-local form: launches of up to 64 rounds (the bench line's em_rounds says how many ran)
        rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 64 * local_launches
    out = {"workload": "human", "genes": 20000, "pairs": pairs,
           "kernel_a_hbm_bytes": int(a_kb * 1024), "em_round_hbm_bytes": int(em_kb * 1024 / rounds), "em_rounds_in_pass": rounds,
           "em_launches_in_pass": local_launches, "finalize_hbm_bytes": int(fin_kb * 1024),
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, bench.py --steps 1 --warmup 0), per launch = "
                     "(FETCH_SIZE + WRITE_SIZE) x 1024 B, raw counters (no x2: the dominant traffic of kernel A is random 64-byte "
                     "bucket lines, for which FETCH_SIZE matched the kernel's own count of 64 B x bucket reads to 2.5 %; see "
                     "profiles/README.md); EM: all launches of the EM kernels / rounds executed (component-local form: the per-launch "
                     "load and store of the groups, nothing per round); finalize: the kernels of kamd_ec_finalize, one step"}
    json.dump(out, open(f"{out_dir}/traffic.json", "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
