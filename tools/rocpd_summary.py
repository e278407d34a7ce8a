#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel calls / total / average duration, like `--stats` prints.
usage: tools/rocpd_summary.py <results.db> [out.csv]"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:<>]+?)(<.*)?\(", name)
    base = name.split("(")[0]
    return base if len(base) < 100 else base[:97] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    out.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
    for n, c, t, a, p in rows:
        out.writerow([short(n), c, round(t, 1), round(a, 3), round(p, 3)])


if __name__ == "__main__":
    main()
