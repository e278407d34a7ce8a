"""kallisto_amd/csrc/kamd_em_local.h on the CPU: the component-local form of EMAlgorithm::run (groups of connected
components iterate on their own; only the stop rule is global and is resolved by speculative chunks + replay) must give
the oracle's number of rounds and its abundances, for any group size and chunk length.  The header is not wired into
kamd_em_run yet; this pins its plan builder, per-group rounds and driver before the device side is written."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from tests import common, emu_binding as E


def _gene_matrix(n_genes, seed, zero_frac=0.05):
    """gene families only (one connected component per gene), some zero counts, singleton rows, transcripts in no row"""
    rng = np.random.default_rng(seed)
    iso = np.minimum(rng.geometric(0.12, n_genes), 40)
    t0 = np.concatenate([[0], np.cumsum(iso)])
    T = int(t0[-1]) + 4
    sets = {}
    for g in range(n_genes):
        for _ in range(rng.poisson(20)):
            m = min(iso[g], max(1, rng.geometric(0.25)))
            r = tuple(sorted(int(x) + int(t0[g]) for x in rng.choice(iso[g], m, replace=False)))
            sets[r] = sets.get(r, 0) + int(rng.pareto(1.2) * 3) + (0 if rng.random() < zero_frac else 1)
    keys = list(sets)
    rng.shuffle(keys)
    off = np.zeros(len(keys) + 1, np.uint64)
    off[1:] = np.cumsum([len(k) for k in keys])
    ids = np.array([t for k in keys for t in k], np.uint32)
    cnt = np.array([sets[k] for k in keys], np.uint32)
    return off, ids, cnt, rng.uniform(150, 3000, T), T


def _run(off, ids, cnt, eff, T, budget=64 * 1024, target=1 << 40, chunk=64, n_iter=10000, min_rounds=50, builder=0):
    L = E.lib()
    alpha = np.zeros(T); abz = np.zeros(T)
    rounds = C.c_int32(0); ng = C.c_uint32(0); mb = C.c_uint64(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    off = np.ascontiguousarray(off, np.uint64); ids = np.ascontiguousarray(ids, np.uint32); cnt = np.ascontiguousarray(cnt, np.uint32)
    eff = np.ascontiguousarray(eff, np.float64)
    rc = L.emu_em_local(p(off), p(ids), p(cnt), C.c_uint64(len(cnt)), p(eff), C.c_uint64(T), C.c_uint64(budget), C.c_uint64(target),
                        int(n_iter), int(min_rounds), int(chunk), p(alpha), p(abz), C.byref(rounds), C.byref(ng), C.byref(mb), int(builder))
    return rc, alpha, abz, rounds.value, ng.value, mb.value


def _check_plan(off, ids, cnt, eff, T, budget, target, builder=0):
    L = E.lib()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    off = np.ascontiguousarray(off, np.uint64); ids = np.ascontiguousarray(ids, np.uint32); cnt = np.ascontiguousarray(cnt, np.uint32)
    eff = np.ascontiguousarray(eff, np.float64)
    return L.emu_em_local_check_plan(p(off), p(ids), p(cnt), C.c_uint64(len(cnt)), p(eff), C.c_uint64(T), C.c_uint64(budget), C.c_uint64(target), int(builder))


@pytest.mark.parametrize("budget,target", [(1 << 30, 1 << 40), (16 * 1024, 1 << 40), (1 << 30, 300), (6 * 1024, 150)])
def test_plan_holds_the_matrix(budget, target):
    off, ids, cnt, eff, T = _gene_matrix(200, 11)
    assert _check_plan(off, ids, cnt, eff, T, budget, target) == 0


@pytest.mark.parametrize("budget,target", [(1 << 30, 1 << 40), (1 << 30, 300), (12 * 1024, 150), (1 << 30, 1)])
def test_plan_built_by_the_data_parallel_steps(budget, target):
    """the steps the device set-up is made of (one index each, plain stores + atomic adds, scans in between), run serially"""
    off, ids, cnt, eff, T = _gene_matrix(200, 11)
    assert _check_plan(off, ids, cnt, eff, T, budget, target, builder=1) == 0
    # a budget that some group cannot meet is reported, not violated
    assert _check_plan(off, ids, cnt, eff, T, 2 * 1024, 1 << 40, builder=1) == -1


@pytest.mark.parametrize("target", [1 << 40, 300, 90])
def test_two_size_classes_of_groups(target):
    """components of at most 40 entries packed into small groups (one wavefront each on the device), the others into groups of
    `target` entries: the plan still holds the matrix, small groups come first and stay small, and the EM is the oracle's"""
    off, ids, cnt, eff, T = _gene_matrix(200, 11)
    assert _check_plan(off, ids, cnt, eff, T, 1 << 30, target, builder=11) == 0
    a_o, abz_o, r_o = O.em_run(off, ids, cnt, eff, T)
    for b in (11, 12, 14):     # CSR rounds; sliced ELLPACK with segments split above 64 / 3 entries
        rc, a, abz, r, ng, _ = _run(off, ids, cnt, eff, T, 1 << 30, target, 16, builder=b)
        assert rc == 0 and r == r_o and ng > 20
        common.assert_abundance_close(a, a_o, "alpha (two size classes)", rel=1e-9)


@pytest.mark.parametrize("budget,target,chunk", [(1 << 30, 1 << 40, 64), (16 * 1024, 1 << 40, 64), (1 << 30, 400, 7), (8 * 1024, 200, 1),
                                                 (1 << 30, 1 << 40, 10000)])
def test_local_em_equals_the_oracle(budget, target, chunk):
    off, ids, cnt, eff, T = _gene_matrix(150, 5)
    tiny = lambda x: np.where(np.abs(x) < 1e-200, 0.0, x)
    a_o, abz_o, r_o = O.em_run(off, ids, cnt, eff, T)
    rc, a, abz, r, ng, mb = _run(off, ids, cnt, eff, T, budget, target, chunk)
    assert rc == 0 and mb <= budget
    if target < 1000:
        assert ng > 10                                     # really many groups
        rc2, a2, abz2, r2, ng2, _ = _run(off, ids, cnt, eff, T, 1 << 30, target, chunk, builder=1)   # the other plan builder
        assert rc2 == 0 and r2 == r_o and ng2 > 10
        common.assert_abundance_close(a2, a_o, "alpha (plan from the data-parallel steps)", rel=1e-9)
    # the sliced-ELLPACK layout of the same groups (what k_em_sell iterates over) through its host model
    for b in (2, 3, 4):   # segments of more than 64 / 16 / 3 entries split over the lanes of a slice
        rc3, a3, abz3, r3, ng3, mb3 = _run(off, ids, cnt, eff, T, 1 << 30, target, chunk, builder=b)
        assert rc3 == 0 and r3 == r_o
        common.assert_abundance_close(a3, a_o, "alpha (sliced ELLPACK)", rel=1e-9)
        common.assert_abundance_close(tiny(abz3), tiny(abz_o), "alpha_before_zeroes (sliced ELLPACK)", rel=1e-9, floor=1e-12)
    assert r == r_o
    common.assert_abundance_close(a, a_o, "alpha", rel=1e-9)
    common.assert_abundance_close(tiny(abz), tiny(abz_o), "alpha_before_zeroes", rel=1e-9, floor=1e-12)


def test_loop_that_runs_out_has_no_final_round():
    off, ids, cnt, eff, T = _gene_matrix(60, 2)
    a_o, abz_o, r_o = O.em_run(off, ids, cnt, eff, T, n_iter=37, min_rounds=50)
    rc, a, abz, r, _, _ = _run(off, ids, cnt, eff, T, chunk=16, n_iter=37, min_rounds=50)
    assert rc == 0 and r == r_o == 37
    common.assert_abundance_close(a, a_o, "alpha", rel=1e-9)


def test_a_component_that_does_not_fit_is_reported():
    """the hub / long-row matrix of the GPU tests is one giant component: not applicable at a workgroup's LDS budget"""
    from tests.test_gpu_parity import _family_csr
    off, ids, cnt, eff, T = _family_csr(400, 7)
    rc, *_ = _run(off, ids, cnt, eff, T, budget=150 * 1024)
    assert rc == 1
    rc, a, abz, r, ng, _ = _run(off, ids, cnt, eff, T, budget=1 << 30)     # with room for it: one group, same answer
    a_o, _, r_o = O.em_run(off, ids, cnt, eff, T)
    assert rc == 0 and ng == 1 and r == r_o
    common.assert_abundance_close(a, a_o, "alpha", rel=1e-9)
    # the same in the sliced-ELLPACK layout: rows with hundreds of transcripts and hub columns are split over the lanes of a
    # slice (segmented combination of the lanes' partial sums)
    for b in (2, 3, 4):
        rc, a, abz, r, ng, _ = _run(off, ids, cnt, eff, T, budget=1 << 30, builder=b)
        assert rc == 0 and ng == 1 and r == r_o
        common.assert_abundance_close(a, a_o, "alpha (sliced ELLPACK)", rel=1e-9)


@pytest.mark.parametrize("case,variant", [("ref_test_pe", "pe"), ("human_pe", "pe"), ("yeast_se", "se"), ("mosaic_pe", "pe")])
def test_local_em_on_the_golden_ec_matrices(case, variant):
    meta, idx_path, r1, r2 = common.load_case(case)
    o = common.parse_variant(meta["variants"][variant])
    exp = common.load_expected(case, variant)
    ix = O.Index(idx_path)
    buf, off, lens = O.pack_reads(common.interleave(r1, r2 if o["paired"] else None))
    res = O.process_reads(ix, O.Opts(o["paired"], o["fld"], o["sd"], o["single_overhang"], o["strand"]), buf, off, lens)
    eoff, eids, ecnt = res.ec_off, res.ec_ids, res.counts
    eff = exp["eff"]
    a_o, abz_o, r_o = O.em_run(eoff, eids, ecnt, eff, len(eff))
    rc, a, abz, r, ng, _ = _run(eoff, eids, ecnt, eff, len(eff), target=64, chunk=32, builder=1)
    assert rc == 0 and r == r_o
    common.assert_abundance_close(a, a_o, "alpha", rel=1e-9)
    rc, a, abz, r, ng, _ = _run(eoff, eids, ecnt, eff, len(eff), target=64, chunk=32, builder=2)
    assert rc == 0 and r == r_o
    common.assert_abundance_close(a, a_o, "alpha (sliced ELLPACK)", rel=1e-9)
    common.assert_abundance_close(a, exp["alpha"], "alpha vs the reference")       # 1e-4, like every other path
