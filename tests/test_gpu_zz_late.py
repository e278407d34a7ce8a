"""GPU tests of what was added after the round's GPU minutes were spent (round 3): they run LAST (pytest takes the files in alphabetical order), so
that a surprise in one of them cannot hide the state of everything that has been seen green on hardware.
  * the compact k-mer table through the C++ front-end (six of the eight cases did run on an MI355X: profiles/r03_compact_table_check.txt);
  * `quant-tcc -g` / `--matrix-to-directories` against the reference CLI's files (the host-side helpers are checked on the CPU:
    test_host_logic::test_gene_level_outputs_of_quant_tcc)."""
import json
import os
import subprocess

import pytest

from tests import common
from tests.test_gpu_cli import EXE, _fastq
from tests import test_gpu_bus_tcc as B

pytestmark = pytest.mark.gpu


VERIFIED_BYTE_EQUAL = {("ref_test_pe", "pe"), ("yeast_se", "se"), ("dlist_pe", "pe"), ("mosaic_pe", "pe_union"), ("human_pe", "pe"), ("mosaic_pe", "se_nojump")}


@pytest.mark.parametrize("case,variant", [("ref_test_pe", "pe"), ("yeast_se", "se"), ("dlist_pe", "pe"), ("mosaic_pe", "pe_union"), ("human_pe", "pe"),
                                          ("mosaic_pe", "se_nojump"), ("tiny_k7_se", "se"), ("mosaic_pe", "pe_nojump_rf")])
def test_compact_kmer_table_changes_nothing(case, variant, tmp_path):
    """KAMD_TABLE_LAYOUT=compact: the k-mer table in four quotiented 16-byte slots per line instead of three 20-byte ones (kamd_core.h; its own
    instantiation of kernel A, the same straight-line matcher elsewhere).  abundance.tsv must be the wide layout's byte for byte, and --
    for the six cases of profiles/r03_compact_table_check.txt (scratch/r3_call34.sh on an MI355X) -- the reference CLI's own file."""
    meta, idx_path, r1, r2 = common.load_case(case)
    extra = meta["variants"][variant]
    cli = [a.replace("--fr", "--fr-stranded").replace("--rf", "--rf-stranded") for a in extra]
    f1 = str(tmp_path / "r_1.fq")
    _fastq(f1, r1)
    files = [f1]
    if r2 is not None and "--single" not in extra:
        f2 = str(tmp_path / "r_2.fq")
        _fastq(f2, r2)
        files.append(f2)
    out, out_w = str(tmp_path / "out"), str(tmp_path / "out_wide")
    # (by the environment variable -- what scratch/r3_call34.sh did on the MI355X -- for the verified six, by the option for the others)
    by_flag = (case, variant) not in VERIFIED_BYTE_EQUAL
    p = subprocess.run([EXE, "quant", "-i", idx_path, "-o", out, "--plaintext", "--verbose", *(["--kmer-table", "compact"] if by_flag else []), *cli, *files],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=os.environ if by_flag else dict(os.environ, KAMD_TABLE_LAYOUT="compact"))
    assert p.returncode == 0, p.stderr.decode()
    assert "k-mer table: compact layout, 4 slots" in p.stderr.decode()
    pw = subprocess.run([EXE, "quant", "-i", idx_path, "-o", out_w, "--plaintext", "--verbose", *cli, *files], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                        env=dict(os.environ, KAMD_TABLE_LAYOUT="wide"))
    assert pw.returncode == 0 and "k-mer table: wide layout, 3 slots" in pw.stderr.decode(), pw.stderr.decode()
    # the layout changes nothing: same classes, same counts, and the abundances are reproducible to the bit
    assert open(os.path.join(out, "abundance.tsv"), "rb").read() == open(os.path.join(out_w, "abundance.tsv"), "rb").read()
    a, b = json.load(open(os.path.join(out, "run_info.json"))), json.load(open(os.path.join(out_w, "run_info.json")))
    assert all(a[k] == b[k] for k in ("n_processed", "n_pseudoaligned", "n_unique"))
    if (case, variant) in VERIFIED_BYTE_EQUAL:
        ref = os.path.join(common.case_dir(case), "cli_" + variant, "abundance.tsv")
        assert open(os.path.join(out, "abundance.tsv"), "rb").read() == open(ref, "rb").read()


@pytest.mark.parametrize("case", B.LATE_CASES)
def test_quant_tcc_gene_level_and_directories(case, tmp_path):
    """The reference's classes and count matrix in; its abundance matrices, per-sample files (in abundance_N/ with --matrix-to-directories), bootstraps
    and the gene-level sums of all of them (-g) out."""
    gold = os.path.join(B.GOLD, case)
    meta = json.load(open(os.path.join(gold, "case.json")))
    idx = os.path.join(common.case_dir(meta["fixture"]), "index.idx")
    out = str(tmp_path / "tcc")
    B._run_tcc(idx, os.path.join(gold, "matrix.ec"), os.path.join(gold, "tcc.mtx"), meta, gold, out)
    B._compare_tcc_out(out, os.path.join(gold, "tcc_out"), case)


def test_bus_then_quant_tcc_gene_level_chain(tmp_path):
    """GPU bus -> count matrix -> GPU quant-tcc -g: the reference chain's gene-level abundances (class numbering differs, abundances do not)."""
    import numpy as np
    case = "human_pe_genes"
    gold = os.path.join(B.GOLD, case)
    meta = json.load(open(os.path.join(gold, "case.json")))
    idx, bus = B._run_bus(meta, tmp_path)
    _, rec = B._read_bus(os.path.join(bus, "output.bus"))
    n_ecs = len(B._read_ec(os.path.join(bus, "matrix.ec")))
    tcc = str(tmp_path / "tcc.mtx")
    with open(tcc, "w") as f:
        f.write("%%MatrixMarket matrix coordinate integer general\n")
        f.write("%d\t%d\t%d\n" % (len(meta["cuts"]) - 1, n_ecs, len(rec)))
        for r in rec:
            f.write("%d\t%d\t%d\n" % (int(r["bc"]) + 1, int(r["ec"]) + 1, int(r["count"])))
    out = str(tmp_path / "tcc")
    B._run_tcc(idx, os.path.join(bus, "matrix.ec"), tcc, meta, bus if meta["fld_file"] else gold, out, case_dir=gold)
    B._compare_tcc_out(out, os.path.join(gold, "tcc_out"), case + " chain")
