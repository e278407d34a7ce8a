"""Shared helpers of the test-suite: golden fixtures (tests/golden/, produced by the unmodified reference through
oracle/_ref/dump_ec -- see tests/golden/make_golden.py) and option parsing."""
from __future__ import annotations

import gzip
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["ref_test_pe", "yeast_se", "human_pe", "tiny_k7_se", "dlist_pe", "mosaic_pe", "stress_pe"]
MAX_FRAG_LEN = 1000


def case_dir(name):
    return os.path.join(GOLDEN, name)


def load_case(name):
    d = case_dir(name)
    with open(os.path.join(d, "case.json")) as f:
        meta = json.load(f)

    def lines(p):
        with gzip.open(p, "rb") as f:
            return [x.rstrip(b"\n") for x in f]
    r1 = lines(os.path.join(d, "reads_1.txt.gz"))
    r2 = lines(os.path.join(d, "reads_2.txt.gz")) if meta["paired"] else None
    return meta, os.path.join(d, "index.idx"), r1, r2


def parse_variant(extra):
    """dump_ec / kallisto quant flags -> dict(paired, fld, sd, single_overhang, strand, no_jump, boot, seed)."""
    o = dict(paired=1, fld=0.0, sd=0.0, single_overhang=0, strand=0, no_jump=0, union=0, boot=0, seed=42)
    it = iter(extra)
    for a in it:
        if a == "--single":
            o["paired"] = 0
        elif a == "-l":
            o["fld"] = float(next(it))
        elif a == "-s":
            o["sd"] = float(next(it))
        elif a == "--single-overhang":
            o["single_overhang"] = 1
        elif a == "--fr":
            o["strand"] = 1
        elif a == "--rf":
            o["strand"] = 2
        elif a == "--no-jump":
            o["no_jump"] = 1
        elif a == "--union":
            o["union"] = 1
        elif a == "--boot":
            o["boot"] = int(next(it))
        elif a == "--seed":
            o["seed"] = int(next(it))
    return o


def load_expected(name, variant):
    res = {"nproc": 0, "ecs": {}, "flens": np.zeros(MAX_FRAG_LEN, np.uint32), "tr": [], "bs": {}}
    with open(os.path.join(case_dir(name), f"expected_{variant}.txt")) as f:
        for line in f:
            t = line.split()
            if t[0] == "NPROC":
                res["nproc"] = int(t[1])
            elif t[0] == "EC":
                res["ecs"][tuple(int(x) for x in t[1].split(","))] = int(t[2])
            elif t[0] == "FLEN":
                res["flens"][int(t[1])] = int(t[2])
            elif t[0] == "TR":
                res["tr"].append((int(t[2]), float(t[3]), float(t[4]), float(t[5])))
            elif t[0] == "BS":
                res["bs"].setdefault(int(t[1]), []).append(float(t[3]))
    res["lens"] = np.array([x[0] for x in res["tr"]], np.int64)
    res["eff"] = np.array([x[1] for x in res["tr"]], np.float64)
    res["alpha"] = np.array([x[2] for x in res["tr"]], np.float64)
    res["abz"] = np.array([x[3] for x in res["tr"]], np.float64)
    return res


def interleave(r1, r2):
    if r2 is None:
        return list(r1)
    return [x for p in zip(r1, r2) for x in p]


def all_variants():
    out = []
    for c in CASES:
        with open(os.path.join(case_dir(c), "case.json")) as f:
            meta = json.load(f)
        for v in meta["variants"]:
            out.append((c, v))
    return out


def assert_abundance_close(got, want, what, rel=1e-4, floor=1e-7):
    """The tolerance BASELINE.json states for estimated abundances: 1e-4 relative; entries below `floor` (absolute,
    in count units) only need to agree to within the floor.  The zero pattern must be identical."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, what
    big = np.abs(want) > floor
    relerr = np.abs(got[big] - want[big]) / np.abs(want[big])
    assert relerr.size == 0 or relerr.max() <= rel, f"{what}: max rel err {relerr.max():.3e}"
    assert np.all(np.abs(got[~big] - want[~big]) <= floor), f"{what}: small entries differ"
    assert np.array_equal(got == 0, want == 0), f"{what}: zero pattern differs"
