"""The line bench.py prints (VERDICT r5 "next round" #1): a *contract line* of at most bench.CONTRACT_MAX_BYTES bytes built from the run's detail
document -- figures only -- with everything else in a file.  The round-5 driver record did not parse because the one line had grown to 24.9 KB;
these tests rebuild the line from recorded detail documents (profiles/) and from a worst-case one, and drive the child-leg plumbing
(run_child / write_detail) with stand-ins.  No GPU."""
from __future__ import annotations

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

RECORDED = ["r05_bench_default_line.json", "r05_bench_stress_full_size.json", "r05_bench_gencode_size.json", "r05_bench_config2_yeast.json",
            "r04_bench_two_ranks_shared_gpu.json", "r04_bench.json", "r03_bench.json"]


@pytest.mark.parametrize("name", RECORDED)
def test_contract_line_from_a_recorded_run(name):
    d = json.load(open(os.path.join(ROOT, "profiles", name)))
    text = bench.contract_line(d, "bench_detail.json")
    assert "\n" not in text and len(text) < 4096 and len(text) < bench.CONTRACT_MAX_BYTES
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["value"] == d["value"] and line["ms_per_step"] == d["ms_per_step"] and line["n_gpus"] == d["n_gpus"]
    assert line["config"]["workload"] and "model" not in line["config"]
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert "traffic" in rf and rf["launch_ms"] > 0
    if "cpu_baseline" in d:
        cb = line["cpu_baseline"]
        assert cb["value"] == pytest.approx(d["cpu_baseline"]["value"]) and cb["kind"] == "reference" and cb["cores"] >= 1 and cb["sample"]
    # no prose: no string in the line is longer than a label
    def strings(x):
        if isinstance(x, dict):
            for v in x.values():
                yield from strings(v)
        elif isinstance(x, list):
            for v in x:
                yield from strings(v)
        elif isinstance(x, str):
            yield x
    assert max(len(s) for s in strings(line)) <= 200


def test_contract_line_of_the_default_run_carries_the_verdicts():
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default_line.json")))
    line = json.loads(bench.contract_line(d))
    assert line["parity"] == {"prefix_ok": True, "tail_ok": True, "full_size_ok": True, "stress_ok": True, "gencode_ok": True,
                              "full_size": {"pairs_or_reads": 30000000, "n_ecs": [618012, 618012], "em_rounds": [1371, 1371], "est_counts_max_rel_err": pytest.approx(1.7e-13, rel=0.01)}}
    assert line["stress"]["value"] == pytest.approx(74.9463) and line["gencode_size"]["ms_per_step"] == pytest.approx(39.306)
    assert line["cpu_baseline"]["pseudoalign_seconds"] == pytest.approx(20.21) and line["cpu_baseline"]["em_seconds"] == pytest.approx(129.48)
    assert line["roofline"]["kernel"] == "k_match_v3" and line["roofline"]["algorithmic_bytes_per_launch"] == 9555141048
    assert line["roofline_em"]["bound"] == "lds" and line["roofline_finalize"]["frac"] == pytest.approx(0.05519)
    assert line["bootstrap"]["replicates"] == 100


def test_contract_line_reports_failed_and_skipped_legs():
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default_line.json")))
    d["parity_check_tail"] = {"ok": False, "error": "x" * 5000}
    d["stress"] = {"skipped": "400 s into the run " + "y" * 3000}
    d["gencode_size"] = {"error": "rc 1: " + "z" * 3000}
    d["parity_check_full_size"] = {"ok": False, "error": "w" * 3000}
    d["cpu_baseline"] = {"value": None, "unit": "M read pairs/s", "cores": 16, "kind": "reference", "sample": "failed: " + "v" * 3000}
    text = bench.contract_line(d)
    line = json.loads(text)
    assert len(text) < 4096
    assert line["parity"]["tail_ok"] is False and line["parity"]["full_size_ok"] is False and line["parity"]["stress_ok"] is None and line["parity"]["gencode_ok"] is None
    assert line["stress"]["skipped"].startswith("400 s") and line["gencode_size"]["skipped"].startswith("rc 1")
    assert line["cpu_baseline"]["value"] is None and line["roofline"]["frac"] > 0


def test_contract_line_is_bounded_whatever_the_detail_holds():
    """a detail document stuffed with long strings and large tables in every block still gives a line under the bound (blocks are dropped, never the
    contract's fields, `roofline` or `cpu_baseline`)"""
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default_line.json")))
    for k, v in list(d.items()):
        if isinstance(v, dict):
            v["note"] = "n" * 20000
            v["table"] = [{"a": i, "b": "t" * 50} for i in range(500)]
            for kk, vv in list(v.items()):
                if isinstance(vv, str):
                    v[kk] = vv + "s" * 4000
    d["breakdown_ms"].update({f"stage_{i}": float(i) for i in range(400)})
    text = bench.contract_line(d)
    line = json.loads(text)
    assert len(text) <= bench.CONTRACT_MAX_BYTES
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0 and line["value"] == d["value"] and "parity" in line


def test_detail_file_and_child_plumbing(tmp_path, monkeypatch):
    """write_detail puts the whole document in a file; run_child hands a child run of the script a --detail-file and returns that document (the child's
    stdout -- the short line -- is not what the parent digests any more)"""
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_config2_yeast.json")))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rel = bench.write_detail(d, str(tmp_path / "detail.json"))
    assert rel == "detail.json" and json.load(open(tmp_path / "detail.json")) == d
    os.makedirs(tmp_path / "gpurun_out")
    bench.write_detail(d, str(tmp_path / "detail.json"))
    assert json.load(open(tmp_path / "gpurun_out" / "detail.json"))["value"] == d["value"]   # a copy where gpurun brings files back from
    child = tmp_path / "child.py"
    child.write_text("import sys, json\n"
                     "p = sys.argv[sys.argv.index('--detail-file') + 1]\n"
                     "json.dump({'value': 3.5, 'config': {'workload': 'w'}, 'roofline': {'frac': 0.25}}, open(p, 'w'))\n"
                     "print('{\"value\": 3.5}')\n")
    got = bench.run_child([sys.executable, str(child)], 60)
    assert got["value"] == 3.5 and bench.bench_line_digest(got)["roofline"]["frac"] == 0.25
    bad = tmp_path / "bad.py"
    bad.write_text("import sys\nsys.stderr.write('boom')\nsys.exit(4)\n")
    with pytest.raises(RuntimeError, match="rc 4"):
        bench.run_child([sys.executable, str(bad)], 60)
    monkeypatch.setattr(bench, "run_child", lambda cmd, t: (_ for _ in ()).throw(RuntimeError("rc 4: boom")))
    assert "rc 4" in bench.stress_leg()["error"] and "rc 4" in bench.config2_leg()["error"] and "rc 4" in bench.gencode_leg(None)["error"]   # a leg never raises


def test_self_launcher_passes_only_the_short_line_through(tmp_path):
    """`python bench.py --gpus N` from a bare shell prints exactly what rank 0 printed: one line, under the bound"""
    stub = tmp_path / "ranks.py"
    stub.write_text("import os, json\n"
                    "if os.environ['RANK'] == '0':\n"
                    "    print('chatter that is not the line')\n"
                    "    print(json.dumps({'metric': 'm', 'value': 1.0, 'n_gpus': int(os.environ['WORLD_SIZE']), 'roofline': {'frac': 0.1}, 'cpu_baseline': {'value': 0.2}}))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["KAMD_BENCH_LAUNCH_SCRIPT"] = str(stub)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2 and json.loads(lines[0])["cpu_baseline"]["value"] == 0.2
