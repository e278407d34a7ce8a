#!/usr/bin/env python
"""Timeline of the last bench step from a rocprofv3 --kernel-trace CSV: kernel, start offset, duration, idle gap before it."""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n).split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, int(r.get("Stream_Id", 0) or 0)))
rows.sort()
# last step = from the last k_match_v3 minus a little (reset memsets) to the end
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_match_v3")]
i0 = starts[-1]
while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 200_000 and not rows[i0 - 1][2].startswith("k_em_sell"):
    i0 -= 1
t0 = rows[i0][0]; prev_end = t0
tot_gap = 0
out = []
for s, e, n, st in rows[i0:]:
    gap = s - prev_end
    if gap > 0: tot_gap += gap
    out.append("%9.1f us  +%8.1f us  gap %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, n[:70]))
    prev_end = max(prev_end, e)
print("\n".join(out if len(out) < 400 else out[:150] + ["..."] + out[-150:]))
print("step span %.3f ms, idle %.3f ms" % ((prev_end - t0) / 1e6, tot_gap / 1e6))
# aggregate gaps by the kernel that follows
agg = {}
prev_end = t0
for s, e, n, st in rows[i0:]:
    g = max(0, s - prev_end); agg[n] = agg.get(n, 0) + g; prev_end = max(prev_end, e)
for n, g in sorted(agg.items(), key=lambda x: -x[1])[:15]:
    print("idle before %-50s %8.1f us" % (n[:50], g / 1e3))
