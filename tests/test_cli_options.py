"""Option checking of the front-end (CheckOptionsEM, src/main.cpp:1600-1805) -- runs without a GPU because every check
happens before the device is touched.  Where the reference binary built by oracle/Makefile is present (the build
container), its messages for the same command line are the expectation; the literal expectations below are what it
printed when this test was written."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "kallisto_amd", "kallisto_amd_quant")
REF = os.path.join(ROOT, "oracle", "_ref", "kallisto")
IDX = os.path.join(ROOT, "tests", "golden", "ref_test_pe", "index.idx")

pytestmark = pytest.mark.skipif(not os.path.exists(EXE), reason="front-end not built (make -C kallisto_amd/csrc all)")


def _errors(exe, args, cwd):
    p = subprocess.run([exe, "quant", *args], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    lines = [l for l in p.stderr.decode().split("\n") if l.startswith("Error:") or l.startswith("       (use")]
    return p.returncode, lines


CASES = [
    (["-o", "o", "a.fq", "b.fq"], ["Error: kallisto index file missing"]),
    (["-i", "/nonexistent.idx", "-o", "o", "a.fq", "b.fq"], ["Error: kallisto index file not found /nonexistent.idx"]),
    (["-i", IDX, "a.fq", "b.fq"], ["Error: need to specify output directory "]),
    (["-i", IDX, "-o", "o"], ["Error: Missing read files"]),
    (["-i", IDX, "-o", "o", "a.fq", "missing.fq"], ["Error: file not found missing.fq"]),
    (["-i", IDX, "-o", "o", "a.fq"], ["Error: paired-end mode requires an even number of input files",
                                     "       (use --single for processing single-end reads)"]),
    (["-i", IDX, "-o", "o", "--single", "a.fq"],
     ["Error: fragment length mean and sd must be supplied for single-end reads using -l and -s"]),
    (["-i", IDX, "-o", "o", "-l", "200", "a.fq", "b.fq"], ["Error: cannot supply mean/sd without supplying both -l and -s"]),
    (["-i", IDX, "-o", "o", "-l", "-5", "-s", "2", "a.fq", "b.fq"], ["Error: invalid value for mean fragment length -5"]),
    (["-i", IDX, "-o", "o", "-l", "5", "-s", "-2", "a.fq", "b.fq"], ["Error: invalid value for fragment length standard deviation -2"]),
    (["-i", IDX, "-o", "o", "-b", "-3", "a.fq", "b.fq"], ["Error: number of bootstrap samples must be a non-negative integer."]),
    (["-i", IDX, "-o", "a.fq", "a.fq", "b.fq"], ["Error: file a.fq exists and is not a directory"]),
    (["-i", IDX, "-o", "o", "-t", "0", "a.fq", "b.fq"], ["Error: invalid number of threads 0"]),
    (["-o", "o", "-l", "3", "a.fq"], ["Error: kallisto index file missing", "Error: paired-end mode requires an even number of input files",
                                      "       (use --single for processing single-end reads)",
                                      "Error: cannot supply mean/sd without supplying both -l and -s"]),
]


@pytest.mark.parametrize("args,expected", CASES)
def test_option_errors_match_the_reference(args, expected, tmp_path):
    for f in ("a.fq", "b.fq"):
        open(tmp_path / f, "w").close()
    rc, got = _errors(EXE, args, str(tmp_path))
    assert rc == 1 and got == expected
    if os.path.exists(REF):
        rrc, ref = _errors(REF, args, str(tmp_path))
        assert rrc == 1 and ref == expected, "the reference binary prints something else now"


@pytest.mark.parametrize("args", [["--bias"], ["--pseudobam"], ["--fusion"], ["--long"], ["--dfk-onlist"]])
def test_options_outside_the_gpu_path_are_refused(args, tmp_path):
    for f in ("a.fq", "b.fq"):
        open(tmp_path / f, "w").close()
    rc, got = _errors(EXE, ["-i", IDX, "-o", "o", *args, "a.fq", "b.fq"], str(tmp_path))
    assert rc == 1 and len(got) == 1 and "outside the GPU quant path" in got[0]


def test_version():
    p = subprocess.run([EXE, "version"], stdout=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout.decode().strip() == "kallisto_amd, compatible with kallisto 0.51.1"


def _sub_errors(exe, sub, args, cwd):
    p = subprocess.run([exe, sub, *args], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, [l for l in p.stderr.decode().split("\n") if l.startswith("Error:")]


BUS_CASES = [
    ("bus", ["-i", IDX, "-o", "o", "a.fq"], ["Error: the technology must be specified via -x, use \"bulk\" for regular RNA-seq reads"]),
    ("bus", ["-x", "bulk", "-o", "o", "a.fq"], ["Error: kallisto index file missing"]),
    ("bus", ["-x", "bulk", "-i", IDX, "-o", "o", "--paired", "a.fq"], ["Error: paired-end mode requires an even number of input files"]),
    ("bus", ["-x", "bulk", "-i", IDX, "-o", "o", "a.fq", "missing.fq"], ["Error: file not found missing.fq"]),
    ("quant-tcc", ["-i", IDX, "-o", "o", "-e", "a.fq"], ["Error: transcript-compatibility counts file missing"]),
    ("quant-tcc", ["-i", IDX, "-o", "o", "-e", "a.fq", "-l", "200", "-s", "20", "-f", "b.fq", "a.fq"],
     ["Error: cannot supply mean or sd while also supplying a fragment length distribution file"]),
    ("quant-tcc", ["-i", IDX, "-o", "o", "-e", "missing.ec", "a.fq"], ["Error: equivalence class file not found missing.ec"]),
]


@pytest.mark.parametrize("sub,args,expected", BUS_CASES)
def test_bus_and_quant_tcc_option_errors(sub, args, expected, tmp_path):
    """CheckOptionsBus (the `-x bulk` branch, src/main.cpp:1048-1107) / CheckOptionsTCCQuant (src/main.cpp:1807-1960)."""
    for f in ("a.fq", "b.fq"):
        open(tmp_path / f, "w").close()
    rc, got = _sub_errors(EXE, sub, args, str(tmp_path))
    assert rc == 1 and got == expected
    if os.path.exists(REF):
        rrc, ref = _sub_errors(REF, sub, args, str(tmp_path))
        assert rrc == 1 and ref == expected, "the reference binary prints something else now"


def test_single_cell_technologies_are_refused(tmp_path):
    open(tmp_path / "a.fq", "w").close()
    rc, got = _sub_errors(EXE, "bus", ["-x", "10xv3", "-i", IDX, "-o", "o", "a.fq"], str(tmp_path))
    assert rc == 1 and got == ["Error: only `-x bulk` runs on the GPU path; single-cell technologies stay with the reference kallisto"]


def test_bench_and_its_side_leg_parse():
    """bench.py's flags (the driver runs it without any) and the side leg's script are at least importable / parseable on a box without a GPU."""
    import ast
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"--table-layout" in p.stdout and b"--no-compact-leg" in p.stdout and b"--gpus" in p.stdout
    ast.parse(open(os.path.join(root, "tools", "compact_table_leg.py")).read())
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "compact_table_leg.py"), "--help"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"--loads" in p.stdout


def test_bench_compact_leg_glue(tmp_path):
    """bench.compact_table_leg with a stand-in for the child: the legs come back with `identical_to_headline`, a failing child becomes an error entry,
    the hand-over directory is removed either way."""
    import sys
    import types
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    res = types.SimpleNamespace(n_pseudoaligned=9, n_unique=4, flens=np.arange(1000, dtype=np.uint32), est_counts=np.array([1.5, 0.0, 2.25]), em_rounds=52)
    fake = tmp_path / "child.py"
    fake.write_text("import sys, json, os, numpy as np\n"
                    "d = sys.argv[sys.argv.index('--dir') + 1]\n"
                    "assert os.path.exists(os.path.join(d, 'words.i32'))\n"
                    "for i, ec in enumerate(([1.5, 0.0, 2.25], [1.5, 0.0, 2.5])):\n"
                    "    np.savez(os.path.join(d, 'result_%d.npz' % i), n_pseudoaligned=9, n_unique=4, flens=np.arange(1000, dtype=np.uint32), est_counts=np.array(ec), em_rounds=52)\n"
                    "print('noise')\n"
                    "print(json.dumps([{'layout': 'compact', 'load': 0.6}, {'layout': 'compact', 'load': 0.5}, {'layout': 'compact', 'error': 'x'}]))\n")
    d = tmp_path / "hand"
    d.mkdir(); (d / "words.i32").write_bytes(b"\0" * 8)
    legs = bench.compact_table_leg("idx", str(d), 10, 100, True, 2, 1, 0, res, script=str(fake))
    assert [e.get("identical_to_headline") for e in legs] == [True, False, None] and not d.exists()
    d.mkdir()
    bad = tmp_path / "bad.py"
    bad.write_text("import sys\nsys.stderr.write('boom')\nsys.exit(3)\n")
    legs = bench.compact_table_leg("idx", str(d), 10, 100, True, 2, 1, 0, res, script=str(bad))
    assert len(legs) == 1 and "rc 3" in legs[0]["error"] and "boom" in legs[0]["error"] and not d.exists()


def test_bench_line_digest_on_a_kept_line():
    """The digest bench.py makes of its child run of config #2, applied to a line kept under profiles/."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    d = json.load(open(os.path.join(root, "profiles", "r03_bench_config2_yeast.json")))
    k = bench.bench_line_digest(d)
    assert k["value"] == d["value"] and k["unit"] == "M reads/s" and "config #2" in k["workload"]
    assert k["roofline"]["frac"] == d["roofline"]["frac"] and k["cpu_baseline"]["kind"] == "reference"
    assert k["parity_check"]["ec_multiset_equal"] is True and k["parity_check_tail"]["ok"] is True
    assert len(json.dumps(k)) < 4000


@pytest.mark.parametrize("form", ["realistic", "regular"])
def test_fastq_spool_writes_what_write_fastq_fast_writes(tmp_path, monkeypatch, form):
    """bench.FastqSpool (the whole input of a run as FASTQ, text assembled chunk by chunk where the reads are generated) against
    bench.write_fastq_fast on the same reads: byte-identical files, record names continue over the chunks -- in the default form (records as a
    sequencer writes them: variable-length Illumina headers, quality strings with a 3' tail; kallisto_amd/synth_fastq.py) and in the fixed-size
    form of rounds 1-4.  The realistic records are read back by the oracle's FASTQ reader: four lines each, both mates named alike."""
    import numpy as np
    import torch
    import bench
    monkeypatch.setenv("KAMD_BENCH_FASTQ", form)
    rng = np.random.default_rng(3)
    L = 37
    r1 = rng.choice(np.frombuffer(b"ACGTN", np.uint8), (1000, L))
    r2 = rng.choice(np.frombuffer(b"ACGTN", np.uint8), (1000, L))
    sp = bench.FastqSpool(str(tmp_path / "spool"), True, L)
    for a, b in ((0, 300), (300, 301), (301, 1000)):
        sp.add([torch.from_numpy(r1[a:b].copy()), torch.from_numpy(r2[a:b].copy())])
    f1, f2 = sp.close()
    bench.write_fastq_fast(str(tmp_path / "w1.fq"), r1)
    bench.write_fastq_fast(str(tmp_path / "w2.fq"), r2, 1)
    assert open(f1, "rb").read() == open(tmp_path / "w1.fq", "rb").read()
    assert open(f2, "rb").read() == open(tmp_path / "w2.fq", "rb").read()
    assert sp.n == 1000
    lines1, lines2 = open(f1, "rb").read().split(b"\n"), open(f2, "rb").read().split(b"\n")
    assert len(lines1) == 4001 and lines1[-1] == b""
    assert [bytes(x) for x in r1] == lines1[1::4] and [bytes(x) for x in r2] == lines2[1::4]
    if form == "regular":
        assert os.path.getsize(f1) == 1000 * sp.rec_bytes
    else:
        assert all(h.startswith(b"@A00587:") for h in lines1[0:4000:4]) and len({len(h) for h in lines1[0:4000:4]}) > 1   # variable-length headers
        assert [h.split(b" ")[0] for h in lines1[0:4000:4]] == [h.split(b" ")[0] for h in lines2[0:4000:4]]             # mates named alike
        assert all(len(q) == L for q in lines1[3::4]) and len(set(lines1[3::4])) > 100                                      # real quality strings
    sp.remove()
    assert not os.path.exists(f1)
