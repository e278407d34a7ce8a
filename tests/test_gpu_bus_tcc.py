"""`kallisto_amd_quant bus -x bulk` and `kallisto_amd_quant quant-tcc` against the reference CLI's
own output for the same command lines (tests/golden/bus_tcc/, produced by tests/golden/make_bus_tcc.py with
oracle/_ref/kallisto at -t 1).

The reference's BUS file has one record per read; bulk records of a sample with the same class are identical, and the GPU
front-end writes them collapsed (what `bustools sort` makes of the reference's file).  Class ids are arbitrary on both
sides, so BUS files are compared as sorted (barcode, reads, transcript set) lines."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "kallisto_amd", "kallisto_amd_quant")
GOLD = os.path.join(common.GOLDEN, "bus_tcc")
# (the cases added when the round's GPU minutes were spent -- gene-level output, --matrix-to-directories -- run from tests/test_gpu_zz_late.py,
# behind everything that has been seen green on hardware)
LATE_CASES = ("human_pe_genes", "ref_test_pe_dirs_boot")
CASES = sorted(c for c in os.listdir(GOLD) if c not in LATE_CASES) if os.path.isdir(GOLD) else []
BUS_DTYPE = np.dtype([("bc", "<u8"), ("umi", "<u8"), ("ec", "<i4"), ("count", "<u4"), ("flags", "<u4"), ("pad", "<u4")])


def _fastq(path, reads):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))


def _sample_files(meta, tmp_path):
    _, idx, r1, r2 = common.load_case(meta["fixture"])
    paired = "--paired" in meta["bus_flags"]
    n, cuts, files = len(r1), meta["cuts"], []
    for s in range(len(cuts) - 1):
        a, b = int(round(cuts[s] * n)), int(round(cuts[s + 1] * n))
        f1 = str(tmp_path / ("s%d_1.fq" % s))
        _fastq(f1, r1[a:b])
        files.append(f1)
        if paired:
            f2 = str(tmp_path / ("s%d_2.fq" % s))
            _fastq(f2, r2[a:b])
            files.append(f2)
    return idx, files


def _read_bus(path):
    b = open(path, "rb").read()
    assert b[:4] == b"BUS\0"
    ver, bclen, umilen, tlen = struct.unpack("<IIII", b[4:20])
    assert b[20:20 + tlen] == b"BUS file produced by kallisto"
    assert (len(b) - 20 - tlen) % 32 == 0
    return (ver, bclen, umilen), np.frombuffer(b[20 + tlen:], dtype=BUS_DTYPE)


def _read_ec(path):
    ecs = []
    for i, line in enumerate(open(path)):
        e, trs = line.split()
        assert int(e) == i
        ids = [int(x) for x in trs.split(",")]
        assert ids == sorted(set(ids))
        ecs.append(tuple(ids))
    return ecs


def _run_bus(meta, tmp_path, extra=()):
    idx, files = _sample_files(meta, tmp_path)
    out = str(tmp_path / "bus")
    p = subprocess.run([EXE, "bus", "-x", "bulk", "-i", idx, "-o", out, "-t", "5", "--batch-size", "700", *meta["bus_flags"], *extra, *files],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KAMD_FASTQ_CHUNK="3000"))
    assert p.returncode == 0, p.stderr.decode()
    return idx, out


def _mtx(path, shape=None):
    lines = [l for l in open(path) if not l.startswith("%")]
    r, c, n = (int(x) for x in lines[0].split())
    m = np.zeros((r, c))
    assert len(lines) - 1 == n
    prev = (0, 0)
    for l in lines[1:]:
        i, j, v = l.split()
        assert (int(i), int(j)) > prev                     # row-major, strictly increasing like the reference's writer
        prev = (int(i), int(j))
        m[int(i) - 1, int(j) - 1] = float(v)
    return m


@pytest.mark.parametrize("case", CASES)
def test_bus_bulk_matches_reference(case, tmp_path):
    assert os.path.exists(EXE), "build kallisto_amd_quant with `make -C kallisto_amd/csrc all`"
    gold = os.path.join(GOLD, case)
    meta = json.load(open(os.path.join(gold, "case.json")))
    _, out = _run_bus(meta, tmp_path)
    hdr, rec = _read_bus(os.path.join(out, "output.bus"))
    assert list(hdr) == meta["bus_header"] == [1, 16, 1]
    assert np.all(rec["umi"] == np.uint64(2 ** 64 - 1)) and np.all(rec["flags"] == 0) and np.all(rec["pad"] == 0)
    keys = list(zip(rec["bc"].tolist(), rec["ec"].tolist()))
    assert keys == sorted(set(keys)), "records must be sorted by (barcode, class) and collapsed"
    ecs = _read_ec(os.path.join(out, "matrix.ec"))
    assert len(set(ecs)) == len(ecs) and set(rec["ec"].tolist()) == set(range(len(ecs)))   # every listed class occurs, none twice
    lines = sorted((int(r["bc"]), ecs[int(r["ec"])], int(r["count"])) for r in rec)
    got = ["%d\t%d\t%s" % (bc, n, ",".join(map(str, s))) for bc, s, n in lines]
    want = open(os.path.join(gold, "bus_expected.txt")).read().split("\n")[:-1]
    assert got == want
    assert int(rec["count"].sum()) == meta["n_records_reference"]
    info = json.load(open(os.path.join(out, "run_info.json")))
    for k, v in json.load(open(os.path.join(gold, "run_info.json"))).items():
        assert info[k] == v, k
    for fn in ("matrix.cells", "matrix.sample.barcodes", "transcripts.txt") + (("flens.txt",) if "--paired" in meta["bus_flags"] else ()):
        assert open(os.path.join(out, fn)).read() == open(os.path.join(gold, fn)).read(), fn
    assert os.path.exists(os.path.join(out, "flens.txt")) == ("--paired" in meta["bus_flags"])
    if case == CASES[0]:
        # --bus-per-read: the reference's record stream -- one 32-byte record per pseudoaligned read, count 1 -- whose collapse is the above
        (tmp_path / "per_read").mkdir()
        _, out1 = _run_bus(meta, tmp_path / "per_read", extra=["--bus-per-read"])
        hdr1, rec1 = _read_bus(os.path.join(out1, "output.bus"))
        assert list(hdr1) == list(hdr) and len(rec1) == meta["n_records_reference"] and np.all(rec1["count"] == 1)
        k1 = list(zip(rec1["bc"].tolist(), rec1["ec"].tolist()))
        assert k1 == sorted(k1)
        # class ids are numbered in order of first appearance, which may differ from run to run: compare through matrix.ec
        from collections import Counter
        ecs1 = _read_ec(os.path.join(out1, "matrix.ec"))
        assert sorted(ecs1) == sorted(ecs)
        assert Counter((bc, ecs1[e]) for bc, e in k1) == Counter({(bc, ecs[e]): int(n) for (bc, e), n in zip(keys, rec["count"].tolist())})


def _run_tcc(idx, ec, tcc, meta, gold, out, case_dir=None):
    fld = ["-f", os.path.join(gold, "flens.txt")] if meta["fld_file"] else []
    flags = [os.path.join(case_dir or gold, a[1:]) if a.startswith("@") else a for a in meta["tcc_flags"]]   # "@file": a file of the case's directory (-g genemap)
    p = subprocess.run([EXE, "quant-tcc", "-i", idx, "-e", ec, "-o", out, *fld, *flags, tcc], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()


def _files_below(d):
    return sorted(os.path.relpath(os.path.join(r, f), d) for r, _, fs in os.walk(d) for f in fs)


def _compare_tcc_out(out, ref, what):
    assert _files_below(out) == _files_below(ref)   # (--matrix-to-directories: the per-sample files sit in abundance_N/)
    for fn in ("matrix.abundance.mtx", "matrix.abundance.tpm.mtx", "matrix.abundance.gene.mtx", "matrix.abundance.gene.tpm.mtx"):
        if os.path.exists(os.path.join(ref, fn)):   # (the gene-level pair: only with -g)
            common.assert_abundance_close(_mtx(os.path.join(out, fn)), _mtx(os.path.join(ref, fn)), f"{what} {fn}", rel=1e-4, floor=1e-5)
    for fn in ("matrix.fld.tsv", "transcript_lengths.txt", "transcripts.txt", "genes.txt"):
        if os.path.exists(os.path.join(ref, fn)):
            assert open(os.path.join(out, fn)).read() == open(os.path.join(ref, fn)).read(), fn
    if os.path.exists(os.path.join(ref, "matrix.efflens.mtx")):
        a, b = _mtx(os.path.join(out, "matrix.efflens.mtx")), _mtx(os.path.join(ref, "matrix.efflens.mtx"))
        assert np.array_equal(a, b)                      # printed with 6 digits from bit-identical doubles
    for fn in _files_below(ref):
        if not fn.endswith(".tsv") or fn == "matrix.fld.tsv":
            continue
        rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(out, fn))]
        grows = [l.rstrip("\n").split("\t") for l in open(os.path.join(ref, fn))]
        assert rows[0] == grows[0] and len(rows) == len(grows)
        gene_file = ".gene" in os.path.basename(fn)   # gene_id, gene_name, est_counts, tpm  instead of  target_id, length, eff_length, est_counts, tpm
        for a, b in zip(rows[1:], grows[1:]):
            assert a[:2 if gene_file else 3] == b[:2 if gene_file else 3], (fn, a, b)
        for col, name in ((2, "est_counts"), (3, "tpm")) if gene_file else ((3, "est_counts"), (4, "tpm")):
            common.assert_abundance_close(np.array([float(r[col]) for r in rows[1:]]), np.array([float(r[col]) for r in grows[1:]]),
                                          f"{what} {fn} {name}", rel=1e-4, floor=1e-5)


@pytest.mark.parametrize("case", CASES)
def test_quant_tcc_matches_reference(case, tmp_path):
    """The reference's own classes and count matrix in, its abundance matrices (and per-sample files, bootstraps) out."""
    gold = os.path.join(GOLD, case)
    meta = json.load(open(os.path.join(gold, "case.json")))
    idx = os.path.join(common.case_dir(meta["fixture"]), "index.idx")
    out = str(tmp_path / "tcc")
    _run_tcc(idx, os.path.join(gold, "matrix.ec"), os.path.join(gold, "tcc.mtx"), meta, gold, out)
    _compare_tcc_out(out, os.path.join(gold, "tcc_out"), case)


@pytest.mark.parametrize("case", [c for c in CASES if "boot" not in c])
def test_bus_then_quant_tcc_chain(case, tmp_path):
    """GPU bus -> count matrix -> GPU quant-tcc gives the reference chain's abundances (class numbering differs, the
    abundances do not depend on it; bootstraps do, so the bootstrap case is left to the test above)."""
    gold = os.path.join(GOLD, case)
    meta = json.load(open(os.path.join(gold, "case.json")))
    idx, bus = _run_bus(meta, tmp_path)
    _, rec = _read_bus(os.path.join(bus, "output.bus"))
    n_ecs = len(_read_ec(os.path.join(bus, "matrix.ec")))
    tcc = str(tmp_path / "tcc.mtx")
    with open(tcc, "w") as f:
        f.write("%%MatrixMarket matrix coordinate integer general\n")
        f.write("%d\t%d\t%d\n" % (len(meta["cuts"]) - 1, n_ecs, len(rec)))
        for r in rec:
            f.write("%d\t%d\t%d\n" % (int(r["bc"]) + 1, int(r["ec"]) + 1, int(r["count"])))
    out = str(tmp_path / "tcc")
    meta2 = dict(meta)
    gold2 = gold
    if meta["fld_file"]:                                   # the chain uses its own flens.txt (identical to the reference's, see above)
        gold2 = bus
    _run_tcc(idx, os.path.join(bus, "matrix.ec"), tcc, meta2, gold2, out, case_dir=gold)
    _compare_tcc_out(out, os.path.join(gold, "tcc_out"), case + " chain")


def test_batch_file_and_plain_tcc(tmp_path):
    """-B batch file: lines with the same id share a barcode (batch_id_mapping, src/ProcessReads.h:211-223); a non-matrix TCC file
    ("class count" lines, zero-based) gives abundance.tsv."""
    gold = os.path.join(GOLD, "ref_test_pe_boot")
    meta = json.load(open(os.path.join(gold, "case.json")))
    idx, files = _sample_files(meta, tmp_path)
    bf = str(tmp_path / "batch.txt")
    with open(bf, "w") as f:
        f.write("# id file1 file2\nA %s %s\n\nB %s %s\nA %s %s\n" % tuple(files))
    out = str(tmp_path / "bus")
    p = subprocess.run([EXE, "bus", "-i", idx, "-o", out, "-B", bf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    assert open(os.path.join(out, "matrix.cells")).read() == "A\nB\nA\n"
    assert open(os.path.join(out, "matrix.sample.barcodes")).read() == "A" * 16 + "\n" + "A" * 15 + "C\n" + "A" * 16 + "\n"
    _, rec = _read_bus(os.path.join(out, "output.bus"))
    ecs = _read_ec(os.path.join(out, "matrix.ec"))
    want = {}
    for l in open(os.path.join(gold, "bus_expected.txt")).read().split("\n")[:-1]:
        bc, n, s = l.split("\t")
        k = ({0: 0, 1: 1, 2: 0}[int(bc)], tuple(int(x) for x in s.split(",")))
        want[k] = want.get(k, 0) + int(n)
    got = {(int(r["bc"]), ecs[int(r["ec"])]): int(r["count"]) for r in rec}
    assert got == want
    assert len(open(os.path.join(out, "flens.txt")).read().split("\n")) == 4          # one line per batch line
    # non-matrix TCC: sample B's counts
    tcc = str(tmp_path / "b.tcc")
    with open(tcc, "w") as f:
        for r in rec:
            if int(r["bc"]) == 1:
                f.write("%d\t%d\n" % (int(r["ec"]), int(r["count"])))
    fl = str(tmp_path / "b.flens")
    open(fl, "w").write(open(os.path.join(out, "flens.txt")).read().split("\n")[1] + "\n")
    o2 = str(tmp_path / "tcc")
    p = subprocess.run([EXE, "quant-tcc", "-i", idx, "-e", os.path.join(out, "matrix.ec"), "-o", o2, "-f", fl, tcc], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(o2, "abundance.tsv"))]
    grows = [l.rstrip("\n").split("\t") for l in open(os.path.join(gold, "tcc_out", "abundance_2.tsv"))]
    assert [r[:3] for r in rows] == [r[:3] for r in grows]
    common.assert_abundance_close(np.array([float(r[3]) for r in rows[1:]]), np.array([float(r[3]) for r in grows[1:]]), "plain tcc", rel=1e-4, floor=1e-5)


def test_ec_upload_through_the_abi():
    """kamd_ec_upload / kamd_ec_set_counts: the EM on caller-supplied classes equals the oracle's EMAlgorithm::run, sample after
    sample on one plan."""
    from kallisto_amd import api
    from oracle import oracle
    gold = os.path.join(GOLD, "human_pe_2")
    ecs = _read_ec(os.path.join(gold, "matrix.ec"))
    m = _mtx(os.path.join(gold, "tcc.mtx")).astype(np.uint32)
    off = np.zeros(len(ecs) + 1, np.uint64)
    off[1:] = np.cumsum([len(e) for e in ecs])
    ids = np.array([t for e in ecs for t in e], np.uint32)
    index = api.Index(os.path.join(common.case_dir("human_pe"), "index.idx"))
    T = index.num_targets
    eff = np.linspace(150.0, 900.0, T)
    ctx = api.Context(0)
    ctx.ec_upload(off, ids)
    for s in range(m.shape[0]):
        ctx.ec_set_counts(m[s])
        alpha, abz, rounds = ctx.em_run(eff)
        o_alpha, o_abz, o_rounds = oracle.em_run(off, ids, m[s], eff, T)
        assert rounds == o_rounds
        common.assert_abundance_close(alpha, o_alpha, f"sample {s}", rel=1e-9, floor=1e-12)
        if s:
            assert ctx.profile()["em_plan_cached"] == 1
    with pytest.raises(api.KallistoAmdError):
        ctx.ec_upload(np.array([0, 2], np.uint64), np.array([3, 1], np.uint32))      # not sorted
