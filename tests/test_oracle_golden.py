"""The oracle (oracle/kallisto_oracle.c) against the golden vectors produced by the unmodified reference."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import common


@pytest.fixture(scope="module")
def indices():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = O.Index(common.load_case(name)[1])
        return cache[name]
    return get


@pytest.mark.parametrize("case,variant", common.all_variants())
def test_oracle_matches_reference(case, variant, indices):
    meta, idx_path, r1, r2 = common.load_case(case)
    o = common.parse_variant(meta["variants"][variant])
    exp = common.load_expected(case, variant)
    ix = indices(case)
    assert ix.k == meta["k"]
    assert np.array_equal(ix.target_lens, exp["lens"])
    buf, off, lens = O.pack_reads(common.interleave(r1, r2 if o["paired"] else None))
    res = O.process_reads(ix, O.Opts(o["paired"], o["fld"], o["sd"], o["single_overhang"], o["strand"], o["no_jump"], o["union"]), buf, off, lens)
    assert res.n_processed == exp["nproc"]
    assert res.multiset() == exp["ecs"]                       # bit-exact EC counts
    assert np.array_equal(res.flens, exp["flens"])            # fragment length sample
    mft = O.mean_frag_lens_trunc(res.flens) if o["fld"] == 0.0 else O.trunc_gaussian_fld(o["fld"], o["sd"])
    eff, _ = O.eff_lens(ix.target_lens, mft)
    assert np.array_equal(eff, exp["eff"])                    # FP64, same operation order -> identical
    alpha, abz, _ = O.em_run(res.ec_off, res.ec_ids, res.counts, eff, ix.num_targets)
    common.assert_abundance_close(alpha, exp["alpha"], "alpha", rel=1e-9)
    common.assert_abundance_close(abz, exp["abz"], "alpha_before_zeroes", rel=1e-9, floor=1e-12)
    if o["boot"]:
        seeds = O.bootstrap_seeds(o["seed"], o["boot"])
        for b in range(o["boot"]):
            samp = O.multinomial_sample(res.counts, int(seeds[b]))
            assert samp.sum() == res.counts.sum()
            a, _, _ = O.em_run(res.ec_off, res.ec_ids, samp, eff, ix.num_targets, weight_counts=res.counts)
            common.assert_abundance_close(a, np.array(exp["bs"][b]), f"bootstrap {b}", rel=1e-9)


def test_degenerate_reads(indices):
    ix = indices("human_pe")
    opts = O.Opts(1, 0.0, 0.0, 0, 0)
    for s1, s2 in [(b"", b""), (b"ACGT", b"ACGT"), (b"N" * 100, b"N" * 100), (b"A" * 31, b"C" * 31)]:
        u, n1, n2 = ix.pseudoalign(opts, s1, s2)
        assert u == [] and n1 == 0 and n2 == 0


def test_mt19937_64_seeds():
    # first outputs of std::mt19937_64 seeded with 5489 (the standard's check value is the 10000th: 9981545732273789042)
    s = O.bootstrap_seeds(5489, 10000)
    assert int(s[0]) == 14514284786278117030 and int(s[9999]) == 9981545732273789042
