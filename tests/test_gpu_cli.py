"""The `kallisto quant`-compatible front-end (kallisto_amd/kallisto_amd_quant) against the reference CLI's own output
(`kallisto quant -t 1 --plaintext`, stored under tests/golden/<case>/cli_<variant>/ by make_golden.py)."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "kallisto_amd", "kallisto_amd_quant")


def _fastq(path, reads):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))


def _table(path):
    rows = [l.rstrip("\n").split("\t") for l in open(path)]
    return rows[0], rows[1:]


HOST_PARSED = {("ref_test_pe", "pe_boot"), ("yeast_se", "se"), ("human_pe", "pe"), ("mosaic_pe", "pe_union_fr")}   # also run through the general reader


@pytest.mark.parametrize("case,variant", [("ref_test_pe", "pe"), ("ref_test_pe", "pe_boot"), ("ref_test_pe", "pe_rf"),
                                          ("yeast_se", "se"), ("yeast_se", "se_fr"), ("human_pe", "pe"),
                                          ("human_pe", "pe_l180"), ("tiny_k7_se", "se"), ("dlist_pe", "pe"), ("dlist_pe", "se_rf"),
                                          ("mosaic_pe", "pe_nojump"), ("mosaic_pe", "se_nojump"), ("mosaic_pe", "se"),
                                          ("mosaic_pe", "pe_union"), ("mosaic_pe", "pe_union_fr"), ("mosaic_pe", "pe_nojump_rf"),
                                          ("mosaic_pe", "se_union_overhang")])
def test_cli_matches_reference_cli(case, variant, tmp_path):
    assert os.path.exists(EXE), "build kallisto_amd_quant with `make -C kallisto_amd/csrc all`"
    meta, idx_path, r1, r2 = common.load_case(case)
    extra = meta["variants"][variant]
    cli = [a.replace("--fr", "--fr-stranded").replace("--rf", "--rf-stranded") for a in extra]
    cli = ["-b" if a == "--boot" else a for a in cli]
    f1 = str(tmp_path / "r_1.fq")
    _fastq(f1, r1)
    files = [f1]
    if r2 is not None and "--single" not in extra:
        f2 = str(tmp_path / "r_2.fq")
        _fastq(f2, r2)
        files.append(f2)
    out = str(tmp_path / "out")
    # the device parser with units of ~20 KB of text (dozens of units even for these small files) ...
    p = subprocess.run([EXE, "quant", "-i", idx_path, "-o", out, "--plaintext", "--verbose", "-t", "7", *cli, *files],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KAMD_FQ_UNIT_BYTES="20000", KAMD_FQ_BATCH_ITEMS="700"))
    assert p.returncode == 0, p.stderr.decode()
    assert "device parser: 0 units" not in p.stderr.decode()
    if (case, variant) in HOST_PARSED:
        # ... and the general reader: -t 7 + a tiny chunk size force the parallel memory-mapped FASTQ reader to split the files into many chunks
        out_h = str(tmp_path / "out_host")
        ph = subprocess.run([EXE, "quant", "-i", idx_path, "-o", out_h, "--plaintext", "--batch", "1500", "-t", "7", *cli, *files],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KAMD_FASTQ_CHUNK="3000", KAMD_HOST_PARSE="1"))
        assert ph.returncode == 0, ph.stderr.decode()
        for fn in sorted(os.listdir(out)):
            if fn.endswith(".tsv"):
                assert open(os.path.join(out, fn), "rb").read() == open(os.path.join(out_h, fn), "rb").read(), fn
    gold = os.path.join(common.case_dir(case), "cli_" + variant)
    # run_info.json: same keys in the same order, same values (start_time / call excluded)
    info = json.load(open(os.path.join(out, "run_info.json")))
    keys = list(info)
    assert keys == ["n_targets", "n_bootstraps", "n_processed", "n_pseudoaligned", "n_unique", "p_pseudoaligned", "p_unique",
                    "kallisto_version", "index_version", "k-mer length", "start_time", "call"]
    ginfo = json.load(open(os.path.join(gold, "run_info.json")))
    for k, v in ginfo.items():
        assert info[k] == v, k
    for fn in sorted(os.listdir(gold)):
        if not fn.endswith(".tsv"):
            continue
        # bootstrap replicates are multinomials over the count vector IN EC-ID ORDER: with -b the front-end asks for the
        # reference's -t 1 ids (kamd_ec_track_order), so bs_abundance_N.tsv is compared like abundance.tsv
        h, rows = _table(os.path.join(out, fn))
        gh, grows = _table(os.path.join(gold, fn))
        assert h == gh and len(rows) == len(grows)
        for a, b in zip(rows, grows):
            assert a[:3] == b[:3], (fn, a, b)            # target_id, length, eff_length: identical text
        est = np.array([float(r[3]) for r in rows]); gest = np.array([float(r[3]) for r in grows])
        tpm = np.array([float(r[4]) for r in rows]); gtpm = np.array([float(r[4]) for r in grows])
        common.assert_abundance_close(est, gest, fn + " est_counts", rel=1e-4, floor=1e-5)
        common.assert_abundance_close(tpm, gtpm, fn + " tpm", rel=1e-4, floor=1e-5)
    if case == "human_pe" and variant == "pe":
        # gzip input goes through the serial zlib reader: same result
        import gzip
        gz = []
        for f in files:
            with open(f, "rb") as fi, gzip.open(f + ".gz", "wb") as fo:
                fo.write(fi.read())
            gz.append(f + ".gz")
        out2 = str(tmp_path / "out_gz")
        p2 = subprocess.run([EXE, "quant", "-i", idx_path, "-o", out2, "--plaintext", *cli, *gz], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p2.returncode == 0, p2.stderr.decode()
        assert open(os.path.join(out2, "abundance.tsv")).read() == open(os.path.join(out, "abundance.tsv")).read()
        # ... and through the block-parallel inflate (files this small take the zlib reader unless told otherwise): chunks of 8 KiB of
        # compressed bytes, so that block starts are searched, symbols carry markers and windows are resolved many times over
        out3 = str(tmp_path / "out_pgz")
        p3 = subprocess.run([EXE, "quant", "-i", idx_path, "-o", out3, "--plaintext", "-t", "6", *cli, *gz], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            env=dict(os.environ, KAMD_PARGZIP_MIN_KB="0", KAMD_PARGZIP_CHUNK_KB="8"))
        assert p3.returncode == 0, p3.stderr.decode()
        assert open(os.path.join(out3, "abundance.tsv")).read() == open(os.path.join(out, "abundance.tsv")).read()
    if case == "ref_test_pe" and variant == "pe":
        # BASELINE config #1: the md5 the survey pinned for the reference's abundance.tsv
        md5 = hashlib.md5(open(os.path.join(out, "abundance.tsv"), "rb").read()).hexdigest()
        assert md5 == "0bd5087aba9db4b681073bb84de3fe5f"


H5DUMP = "/opt/conda/bin/h5dump"


@pytest.mark.parametrize("case,variant", [("ref_test_pe", "pe_boot"), ("yeast_se", "se"), ("dlist_pe", "pe")])
def test_cli_writes_the_reference_abundance_h5(case, variant, tmp_path):
    """Without --plaintext the front-end writes abundance.h5 like a USE_HDF5 build of the reference (H5Writer.cpp:4-69): same
    groups / datasets / types / chunking / deflate level / string sizes, same integers and strings, estimated counts and
    bootstrap replicates within 1e-4.  Golden: h5dump of the reference's file (oracle/_ref/kallisto_h5, make_golden.py)."""
    if not os.path.exists(H5DUMP):
        pytest.skip("h5dump not installed")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import h5dump_tools as H
    meta, idx_path, r1, r2 = common.load_case(case)
    extra = meta["variants"][variant]
    cli = [a.replace("--fr", "--fr-stranded").replace("--rf", "--rf-stranded") for a in extra]
    cli = ["-b" if a == "--boot" else a for a in cli]
    f1 = str(tmp_path / "r_1.fq")
    _fastq(f1, r1)
    files = [f1]
    if r2 is not None and "--single" not in extra:
        f2 = str(tmp_path / "r_2.fq")
        _fastq(f2, r2)
        files.append(f2)
    out = str(tmp_path / "out")
    p = subprocess.run([EXE, "quant", "-i", idx_path, "-o", out, "-t", "4", *cli, *files], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    assert os.path.exists(os.path.join(out, "abundance.h5")), p.stderr.decode()
    assert not any(fn.startswith("bs_abundance") for fn in os.listdir(out))        # replicates live in the h5 file
    dump = subprocess.run([H5DUMP, "-p", "-m", "%.12g", os.path.join(out, "abundance.h5")], check=True, stdout=subprocess.PIPE).stdout.decode()
    got = H.parse(dump)
    want = H.parse(open(os.path.join(common.case_dir(case), "cli_" + variant, "abundance.h5.dump")).read())
    assert sorted(got) == sorted(want)
    for name, w in want.items():
        g = got[name]
        if name in ("aux/call", "aux/start_time"):          # run-specific strings: only the layout is comparable
            assert g["type"] == w["type"] and g["dims"] == w["dims"] == 1
            continue
        for key in ("type", "dims", "chunk", "deflate", "strsize"):
            assert g[key] == w[key], (name, key, g[key], w[key])
        if name == "est_counts" or name.startswith("bootstrap/"):
            common.assert_abundance_close(np.array(g["values"]), np.array(w["values"]), name, rel=1e-4, floor=1e-5)
        elif name == "aux/eff_lengths":
            assert np.allclose(g["values"], w["values"], rtol=1e-11, atol=0), name
        else:
            assert g["values"] == w["values"], name


def test_cli_with_a_flattened_index(tmp_path):
    """`flatten` writes the device tables as a file; `quant -i` on that file gives the result of the kallisto index."""
    meta, idx_path, r1, r2 = common.load_case("human_pe")
    f1, f2 = str(tmp_path / "r_1.fq"), str(tmp_path / "r_2.fq")
    _fastq(f1, r1); _fastq(f2, r2)
    flat = str(tmp_path / "index.kamd")
    assert subprocess.run([EXE, "flatten", "-i", idx_path, "-o", flat]).returncode == 0
    outs = []
    for ipath, name in ((idx_path, "a"), (flat, "b")):
        out = str(tmp_path / name)
        p = subprocess.run([EXE, "quant", "-i", ipath, "-o", out, "--plaintext", f1, f2], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        outs.append(_table(os.path.join(out, "abundance.tsv"))[1])
    assert [r[:3] for r in outs[0]] == [r[:3] for r in outs[1]]
    common.assert_abundance_close(np.array([float(r[3]) for r in outs[1]]), np.array([float(r[3]) for r in outs[0]]), "est_counts", rel=1e-9, floor=1e-9)


def test_cli_picks_up_only_its_own_flattened_index(tmp_path):
    """`<index>.kamd` beside an index is used when -- and only when -- the file says it was written from that index (size + hash of head and
    tail, kamd_flat_index_matches).  A .kamd of ANOTHER index dropped beside it with a newer mtime (what `cp -p` / a restored artifact
    leaves behind) must be ignored: the run quantifies against the index that was named.  Both runs leave through the ordinary teardown
    (KAMD_SLOW_EXIT=1: kamd_ctx_destroy / kamd_index_free instead of the front-end's _exit shortcut) and must still exit 0."""
    import shutil
    meta, idx_path, r1, r2 = common.load_case("human_pe")
    other_idx = common.load_case("yeast_se")[1]
    f1, f2 = str(tmp_path / "r_1.fq"), str(tmp_path / "r_2.fq")
    _fastq(f1, r1); _fastq(f2, r2)
    idx = str(tmp_path / "index.idx")
    shutil.copy(idx_path, idx)
    env = dict(os.environ, KAMD_SLOW_EXIT="1")

    def run(name):
        out = str(tmp_path / name)
        p = subprocess.run([EXE, "quant", "-i", idx, "-o", out, "--plaintext", f1, f2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert p.returncode == 0, p.stderr.decode()
        return _table(os.path.join(out, "abundance.tsv"))[1], p.stderr.decode()
    base, err = run("plain")
    assert "flattened tables" not in err
    assert subprocess.run([EXE, "flatten", "-i", idx, "-o", idx + ".kamd"]).returncode == 0
    own, err = run("own")
    assert "using the flattened tables of" in err
    assert [r[:3] for r in own] == [r[:3] for r in base]
    common.assert_abundance_close(np.array([float(r[3]) for r in own]), np.array([float(r[3]) for r in base]), "est_counts", rel=1e-9, floor=1e-9)
    assert subprocess.run([EXE, "flatten", "-i", other_idx, "-o", idx + ".kamd"]).returncode == 0    # a foreign file, newer than the index
    foreign, err = run("foreign")
    assert "ignored" in err and "using the flattened tables of" not in err
    assert [r[:3] for r in foreign] == [r[:3] for r in base]
    common.assert_abundance_close(np.array([float(r[3]) for r in foreign]), np.array([float(r[3]) for r in base]), "est_counts", rel=1e-9, floor=1e-9)
    # --kmer-table compact with a wide .kamd beside the index: the compact table is built from the index itself
    assert subprocess.run([EXE, "flatten", "-i", idx, "-o", idx + ".kamd", "--kmer-table", "wide"]).returncode == 0
    out = str(tmp_path / "compact")
    p = subprocess.run([EXE, "quant", "-i", idx, "-o", out, "--plaintext", "--verbose", "--kmer-table", "compact", f1, f2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr.decode()
    assert "was asked for" in p.stderr.decode() and "ignored" in p.stderr.decode()
    rows = _table(os.path.join(out, "abundance.tsv"))[1]
    assert [r[:3] for r in rows] == [r[:3] for r in base]


@pytest.mark.parametrize("case,variant", [("human_pe", "pe"), ("human_pe", "pe_l180"), ("ref_test_pe", "pe_boot"), ("yeast_se", "se"), ("mosaic_pe", "pe_union"),
                                          ("mosaic_pe", "pe_nojump_rf"), ("mosaic_pe", "se_union_overhang"), ("dlist_pe", "pe")])
def test_cli_several_ranks_on_one_device(case, variant, tmp_path):
    """`--gpus 2 --share-device`: the several-GPU flow of the front-end (one context, pipeline and host thread per rank; units dealt
    round the ranks, to rank 0 while the fragment-length sample is open; kamd_ec_allreduce; kamd_em_run_comm; replicates dealt round
    the ranks) executes on a single-GPU box with host-staged collectives, and must reproduce the reference CLI's files
    (MasterProcessor::update, src/ProcessReads.cpp:424-481, is the merge it replaces)."""
    meta, idx_path, r1, r2 = common.load_case(case)
    extra = meta["variants"][variant]
    cli = [a.replace("--fr", "--fr-stranded").replace("--rf", "--rf-stranded") for a in extra]
    cli = ["-b" if a == "--boot" else a for a in cli]
    f1 = str(tmp_path / "r_1.fq")
    _fastq(f1, r1)
    files = [f1]
    if r2 is not None and "--single" not in extra:
        f2 = str(tmp_path / "r_2.fq")
        _fastq(f2, r2)
        files.append(f2)
    gold = os.path.join(common.case_dir(case), "cli_" + variant)
    ginfo = json.load(open(os.path.join(gold, "run_info.json")))
    for mode, env in (("device", dict(KAMD_FQ_UNIT_BYTES="15000", KAMD_FQ_BATCH_ITEMS="500")), ("host", dict(KAMD_HOST_PARSE="1", KAMD_FASTQ_CHUNK="3000"))):
        out = str(tmp_path / ("out_" + mode))
        p = subprocess.run([EXE, "quant", "-i", idx_path, "-o", out, "--plaintext", "--verbose", "--gpus", "3", "--share-device", "--batch", "1500", "-t", "6",
                            *cli, *files], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr.decode()
        assert "merging the equivalence classes of 3 ranks: host-staged callbacks" in p.stderr.decode()
        info = json.load(open(os.path.join(out, "run_info.json")))
        for k, v in ginfo.items():
            assert info[k] == v, (mode, k)
        h, rows = _table(os.path.join(out, "abundance.tsv"))
        gh, grows = _table(os.path.join(gold, "abundance.tsv"))
        assert h == gh and [r[:3] for r in rows] == [r[:3] for r in grows]     # eff_length: the fragment-length sample is rank 0's, in input order
        common.assert_abundance_close(np.array([float(r[3]) for r in rows]), np.array([float(r[3]) for r in grows]), mode + " est_counts", rel=1e-4, floor=1e-5)
        common.assert_abundance_close(np.array([float(r[4]) for r in rows]), np.array([float(r[4]) for r in grows]), mode + " tpm", rel=1e-4, floor=1e-5)
        if "-b" in cli:
            # EC ids of a merged result have no input order (like the reference at -t > 1), so the replicates are other draws than the
            # golden ones: every replicate file must be there, complete, and sum to the number of pseudoaligned reads
            nb = int(cli[cli.index("-b") + 1])
            for b in range(nb):
                _, brows = _table(os.path.join(out, f"bs_abundance_{b}.tsv"))
                assert len(brows) == len(grows)
                assert abs(sum(float(r[3]) for r in brows) - ginfo["n_pseudoaligned"]) < 1e-6 * ginfo["n_pseudoaligned"] + 1e-3


BOUND = os.path.join(ROOT, "oracle", "_ref", "quant_bound")


@pytest.mark.parametrize("case,variant", [("ref_test_pe", "pe"), ("yeast_se", "se")])
def test_reference_reader_and_writers_bound_to_the_library(case, variant, tmp_path):
    """INTEGRATION.md, compiled and run: oracle/_ref/quant_bound is the REFERENCE's FastqSequenceReader, option struct and
    plaintext writers (linked from the reference's own objects) with the four seams replaced by calls into libkallisto_amd.so
    (oracle/ref_harness/quant_bound.cpp).  Its abundance.tsv must be the reference's: byte for byte on BASELINE config #1 (the md5
    the survey pinned), within 1e-4 on the single-end case."""
    if not os.path.exists(BOUND):
        pytest.skip("oracle/_ref/quant_bound not built (make -C oracle ref_bound needs /root/reference)")
    meta, idx_path, r1, r2 = common.load_case(case)
    extra = meta["variants"][variant]
    f1 = str(tmp_path / "r_1.fq")
    _fastq(f1, r1)
    files = [f1]
    if r2 is not None and "--single" not in extra:
        f2 = str(tmp_path / "r_2.fq")
        _fastq(f2, r2)
        files.append(f2)
    out = str(tmp_path / "out")
    p = subprocess.run([BOUND, "quant", "-i", idx_path, "-o", out, "-t", "4", *extra, *files], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    gold = os.path.join(common.case_dir(case), "cli_" + variant)
    if case == "ref_test_pe":
        assert hashlib.md5(open(os.path.join(out, "abundance.tsv"), "rb").read()).hexdigest() == "0bd5087aba9db4b681073bb84de3fe5f"
    h, rows = _table(os.path.join(out, "abundance.tsv"))
    gh, grows = _table(os.path.join(gold, "abundance.tsv"))
    assert h == gh and len(rows) == len(grows)
    for a, b in zip(rows, grows):
        assert a[:3] == b[:3], (a, b)
    est = np.array([float(r[3]) for r in rows]); gest = np.array([float(r[3]) for r in grows])
    common.assert_abundance_close(est, gest, "est_counts", rel=1e-4, floor=1e-5)
    info, ginfo = json.load(open(os.path.join(out, "run_info.json"))), json.load(open(os.path.join(gold, "run_info.json")))
    for k in ("n_targets", "n_processed", "n_pseudoaligned", "n_unique", "p_pseudoaligned", "p_unique", "index_version", "k-mer length"):
        assert info[k] == ginfo[k], k
