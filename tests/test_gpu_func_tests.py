"""The reference's own functional tests for `kallisto quant` (/root/reference/func_tests/runtests.sh:265-304), replayed
through the C++ front-end kallisto_amd_quant: toy indices with k = 5 / 7 / 11 built by the reference binary, reads of 8-23
bases (shorter than k in places), lower-case and non-ACGT bases, FASTQ headers with a stray leading character, several pairs
of input files, a 70 000-member gzip file on 12 threads.  abundance.tsv must have the md5 the reference's script demands.
Fixtures: tests/golden/func_tests (make_func_tests.sh regenerates them with oracle/_ref/kallisto)."""
import hashlib
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "func_tests")
EXE = os.path.join(ROOT, "kallisto_amd", "kallisto_amd_quant")
CASES = json.load(open(os.path.join(FX, "cases.json")))["cases"]


@pytest.fixture(scope="module")
def large(tmp_path_factory):
    p = tmp_path_factory.mktemp("func") / "large.fastq.gz"
    small = open(os.path.join(FX, "small.fastq.gz"), "rb").read()
    with open(p, "wb") as f:
        f.write(small * 70000)          # runtests.sh:136-140
    return str(p)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reference_func_test_md5(case, large, tmp_path):
    assert os.path.exists(EXE), "build kallisto_amd_quant with `make -C kallisto_amd/csrc all`"
    files = [large if f == "large.fastq.gz" else os.path.join(FX, f) for f in case["files"]]
    out = str(tmp_path / "out")
    cmd = [EXE, "quant", "-o", out, "-i", os.path.join(FX, case["index"]), "--plaintext", *case["args"], *files]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    got = hashlib.md5(open(os.path.join(out, "abundance.tsv"), "rb").read()).hexdigest()
    assert got == case["md5"], (case["name"], open(os.path.join(out, "abundance.tsv")).read()[:600])
