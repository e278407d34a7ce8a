"""bench.FullSizeParity without a GPU: the oracle's result of a golden case stands in for the HIP path's, the unmodified reference runs multi-
threaded with the fragment-length sample handed over (`dump_ec quant -t 4 --flens`), and the report must come out green -- and red when the
stand-in is tampered with.  Pins the harness switches (`--no-em`, `--flens`), the report's logic and, once more, the oracle against the
reference at more than one thread (EC counts do not depend on the thread count; the reference's own sample does)."""
import os
import types

import numpy as np
import pytest

from oracle import oracle as O
from tests import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "dump_ec")), reason="oracle/_ref/dump_ec not built")


def _fastq(path, reads):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")


def _stand_in(case):
    meta, idx_path, r1, r2 = common.load_case(case)
    oix = O.Index(idx_path)
    buf, off, ln = O.pack_reads(common.interleave(r1, r2))
    ores = O.process_reads(oix, O.Opts(1, 0.0, 0.0, 0, 0), buf, off, ln)
    eff, _ = O.eff_lens(oix.target_lens, O.mean_frag_lens_trunc(ores.flens))
    alpha, abz, rounds = O.em_run(ores.ec_off, ores.ec_ids, ores.counts, eff, oix.num_targets)
    ecs = types.SimpleNamespace(ec_off=np.asarray(ores.ec_off), ec_ids=np.asarray(ores.ec_ids), counts=np.asarray(ores.counts), multiset=ores.multiset)
    res = types.SimpleNamespace(ecs=ecs, flens=np.asarray(ores.flens), eff_lens=np.asarray(eff), est_counts=np.asarray(alpha), alpha_before_zeroes=np.asarray(abz),
                                em_rounds=int(rounds), n_processed=len(r1), n_pseudoaligned=int(np.asarray(ores.counts).sum()))
    return idx_path, r1, r2, res


def test_full_size_parity_report_green_and_red(tmp_path, monkeypatch):
    import bench
    monkeypatch.setattr(bench, "CACHE", str(tmp_path / "cache"))
    idx_path, r1, r2, res = _stand_in("ref_test_pe")
    f1, f2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    _fastq(f1, r1); _fastq(f2, r2)
    rep = bench.FullSizeParity(idx_path, [f1, f2], res, 4).finish(timeout_s=300)
    assert rep["ok"], rep
    assert rep["ec_multiset_equal"] and rep["eff_length_equal"] and rep["em_rounds"][0] == rep["em_rounds"][1] and rep["n_processed_ref"] == len(r1)
    assert rep["oracle_em_on_the_gpus_ecs"]["ok"] and rep["est_counts_max_rel_err_tpm_ge_1e-3"] <= 1e-9
    # a result that lost one read of one class must be caught
    bad = types.SimpleNamespace(**vars(res))
    counts = res.ecs.counts.copy(); counts[int(np.argmax(counts))] -= 1
    bad.ecs = types.SimpleNamespace(ec_off=res.ecs.ec_off, ec_ids=res.ecs.ec_ids, counts=counts,
                                    multiset=lambda: {tuple(res.ecs.ec_ids[res.ecs.ec_off[i]:res.ecs.ec_off[i + 1]].tolist()): int(counts[i]) for i in range(len(counts))})
    rep = bench.FullSizeParity(idx_path, [f1, f2], bad, 4).finish(timeout_s=300)
    assert not rep["ok"] and not rep["ec_multiset_equal"]


def test_reference_harness_switches(tmp_path):
    """--no-em stops behind ProcessReads with the same EC multiset at any thread count; --flens makes a threaded run's effective lengths
    those of the -t 1 run"""
    idx_path, r1, r2, res = _stand_in("human_pe")
    f1, f2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    _fastq(f1, r1); _fastq(f2, r2)
    one = O.ref_dump_quant(idx_path, [f1, f2], threads=1)
    many = O.ref_dump_quant(idx_path, [f1, f2], threads=4, no_em=True)
    assert many["ecs"] == one["ecs"] == res.ecs.multiset() and many["tr"] == [] and many["nproc"] == one["nproc"]
    with_flens = O.ref_dump_quant(idx_path, [f1, f2], threads=4, flens=one["flens"])
    assert [t[1] for t in with_flens["tr"]] == [t[1] for t in one["tr"]]          # effective lengths, bit for bit
    assert with_flens["rounds"] == one["rounds"] == res.em_rounds
    assert np.allclose([t[2] for t in with_flens["tr"]], [t[2] for t in one["tr"]], rtol=1e-9, atol=1e-12)
