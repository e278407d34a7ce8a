"""The oracle against the reference's `bus -x bulk` / `quant-tcc` output (tests/golden/bus_tcc, made by make_bus_tcc.py with
oracle/_ref/kallisto).  Runs without a GPU.

It pins the two statements the GPU front-end relies on: (1) in bulk mode BUSProcessor::processBuffer
(src/ProcessReads.cpp:1380-1832) is the quant path with `single_overhang` forced on and the fragment-length sample taken per
sample -- so the oracle's process_reads, sample by sample, must give the reference's BUS records and flens.txt; (2) quant-tcc
(src/main.cpp:2802-3220) is EMAlgorithm::run(10000, 50) per matrix row on effective lengths from that row's fragment-length
distribution (or -l/-s, or all 1) -- so the oracle's EM must give the reference's abundance matrices."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import common

GOLD = os.path.join(common.GOLDEN, "bus_tcc")
CASES = sorted(os.listdir(GOLD)) if os.path.isdir(GOLD) else []


def _opts(meta):
    f = meta["bus_flags"]
    paired = 1 if "--paired" in f else 0
    strand = 1 if "--fr-stranded" in f else 2 if "--rf-stranded" in f else 0
    # single-end: match(partial) + no position filter == quant --single --single-overhang; the mean / sd are not read by anything
    return O.Opts(paired, 0.0 if paired else 200.0, 0.0 if paired else 20.0, 1, strand, 1 if "--no-jump" in f else 0, 1 if "--union" in f else 0)


def _mtx(path):
    lines = [l for l in open(path) if not l.startswith("%")]
    r, c, n = (int(x) for x in lines[0].split())
    m = np.zeros((r, c))
    for l in lines[1:]:
        i, j, v = l.split()
        m[int(i) - 1, int(j) - 1] = float(v)
    return m


@pytest.mark.parametrize("case", CASES)
def test_bulk_bus_is_the_quant_path_per_sample(case):
    gold = os.path.join(GOLD, case)
    meta = json.load(open(os.path.join(gold, "case.json")))
    _, idx_path, r1, r2 = common.load_case(meta["fixture"])
    opts = _opts(meta)
    ix = O.Index(idx_path)
    n, cuts = len(r1), meta["cuts"]
    got, flens_lines = {}, []
    n_proc = n_aln = n_uniq = 0
    for s in range(len(cuts) - 1):
        a, b = int(round(cuts[s] * n)), int(round(cuts[s + 1] * n))
        reads = common.interleave(r1[a:b], r2[a:b] if opts.paired else None)
        buf, off, lens = O.pack_reads(reads)
        res = O.process_reads(ix, opts, buf, off, lens)
        n_proc += res.n_processed
        for tr, cnt in res.multiset().items():
            if cnt:
                got[(s, tr)] = got.get((s, tr), 0) + cnt
                n_aln += cnt
                n_uniq += cnt if len(tr) == 1 else 0
        flens_lines.append(" ".join(str(int(x)) for x in res.flens))
    want = {}
    for l in open(os.path.join(gold, "bus_expected.txt")).read().split("\n")[:-1]:
        bc, cnt, trs = l.split("\t")
        want[(int(bc), tuple(int(x) for x in trs.split(",")))] = int(cnt)
    assert got == want
    info = json.load(open(os.path.join(gold, "run_info.json")))
    assert (n_proc, n_aln, n_uniq) == (info["n_processed"], info["n_pseudoaligned"], info["n_unique"])
    if opts.paired:
        assert "\n".join(flens_lines) + "\n" == open(os.path.join(gold, "flens.txt")).read()


@pytest.mark.parametrize("case", CASES)
def test_quant_tcc_is_the_em_per_row(case):
    gold = os.path.join(GOLD, case)
    meta = json.load(open(os.path.join(gold, "case.json")))
    idx_path = os.path.join(common.case_dir(meta["fixture"]), "index.idx")
    ix = O.Index(idx_path)
    T = ix.num_targets
    # the D-list fixture: quant-tcc drops the D-list's pseudo-targets (index.load(opt, false, false)); its matrices have the on-list count
    ab_ref = _mtx(os.path.join(gold, "tcc_out", "matrix.abundance.mtx"))
    T_on = ab_ref.shape[1]
    assert T_on <= T
    ecs = []
    for i, line in enumerate(open(os.path.join(gold, "matrix.ec"))):
        e, trs = line.split()
        assert int(e) == i
        ecs.append(sorted(int(x) for x in trs.split(",")))
    off = np.zeros(len(ecs) + 1, np.uint64)
    off[1:] = np.cumsum([len(e) for e in ecs])
    ids = np.array([t for e in ecs for t in e], np.uint32)
    tcc = _mtx(os.path.join(gold, "tcc.mtx")).astype(np.uint32)
    flags = meta["tcc_flags"]
    for s in range(tcc.shape[0]):
        if meta["fld_file"]:
            fl = np.array(open(os.path.join(gold, "flens.txt")).read().split("\n")[s].split(), np.uint32)
            eff, _ = O.eff_lens(ix.target_lens[:T_on], O.mean_frag_lens_trunc(fl))
        elif "-l" in flags:
            eff, _ = O.eff_lens(ix.target_lens[:T_on], O.trunc_gaussian_fld(float(flags[flags.index("-l") + 1]), float(flags[flags.index("-s") + 1])))
        else:
            eff = np.ones(T_on)
        alpha, _, _ = O.em_run(off, ids, tcc[s], eff, T_on)
        common.assert_abundance_close(alpha, ab_ref[s], f"{case} sample {s} abundance", rel=1e-4, floor=1e-5)
        tpm = O.counts_to_tpm(alpha, eff)
        common.assert_abundance_close(tpm, _mtx(os.path.join(gold, "tcc_out", "matrix.abundance.tpm.mtx"))[s], f"{case} sample {s} tpm", rel=1e-4, floor=1e-5)
        if os.path.exists(os.path.join(gold, "tcc_out", "matrix.efflens.mtx")):
            el = _mtx(os.path.join(gold, "tcc_out", "matrix.efflens.mtx"))[s]
            nz = alpha > 0
            assert np.allclose(el[nz], eff[nz], rtol=1e-5)     # the file prints 6 digits, and only where the abundance is non-zero
