"""BASELINE.json's full size (config #3: human-sized index, 30 M PE-100 pairs on one GPU) through size-independent
properties -- the oracle cannot run at this size in test time, so the HIP path is checked against itself and against
conservation laws of the algorithm:
  * batching invariance: one 30 M batch == several ragged batches (EC multiset identical);
  * sharding linearity: multiset(shard A) + multiset(shard B) == multiset(A u B)  (what the multi-GPU merge relies on);
  * conservation: sum of EC counts == pairs reported pseudoaligned; the EM conserves mass (sum alpha == sum counts),
    TPM sums to 1e6, a bootstrap resample keeps N;
  * a 20 k-pair prefix against the oracle, bit-exact;
  * a 200 k-pair prefix against the UNMODIFIED REFERENCE at -t 1 (oracle/_ref/dump_ec, prebuilt, travels with the snapshot):
    EC multiset, flens and eff_length identical, est_counts / TPM within 1e-4 -- the tolerance BASELINE.json states;
  * ALL 30 M pairs against the unmodified reference on all cores (bench.FullSizeParity): the EC multiset of the whole run identical, the
    reference's EM on its 618 k ECs stops in the same round and agrees within 1e-4, the oracle's EM on the GPU's own ECs within 1e-9;
  * the last 200 k pairs as the final batch of a run over all of them against the reference (bench.tail_parity).
The index is built by the reference binary (oracle/_ref/kallisto, travels with the repo); skipped when it is absent."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def world():
    import torch
    import bench
    import kallisto_amd as ka
    from kallisto_amd.synth_gpu import ReadSimulator
    if not os.path.exists(bench.REF_BIN):
        pytest.skip("oracle/_ref/kallisto not built: cannot create the human-sized index")
    cat, tlens, idx_path = bench.prepare_workload("human", 20000, True)
    index = ka.Index(idx_path)
    ctx = ka.Context(0)
    ctx.upload(index)
    dev = torch.device("cuda", 0)
    L, n = 100, 30_000_000
    sim = ReadSimulator(cat, tlens, dev, seed=4242, read_len=L)
    rec = ka.packed_record_words(L)
    words = torch.empty(n * 2 * rec, dtype=torch.int32, device=dev)
    lens = torch.empty(n * 2, dtype=torch.int16, device=dev)
    prefix = None
    # the whole input as FASTQ too (13 GB), when the reference harness is here and the disk has room: test_whole_run_against_reference
    import shutil
    spool = None
    os.makedirs(bench.CACHE, exist_ok=True)
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "dump_ec")) and shutil.disk_usage(bench.CACHE).free > n * 2 * 216 + (8 << 30):
        spool = bench.FastqSpool(os.path.join(bench.CACHE, f"fullsize_test_{os.getpid()}"), True, L)
    for s in range(0, n, 2_000_000):
        r1, r2 = sim.draw(2_000_000)
        if spool is not None:
            spool.add([r1, r2])
        if s == 0:
            prefix = (r1[:200_000].cpu().numpy(), r2[:200_000].cpu().numpy())
        w, l = ctx.pack_reads(torch.stack([r1, r2], 1).reshape(-1, L), L)
        words[s * 2 * rec:(s + 2_000_000) * 2 * rec] = w
        lens[2 * s:2 * (s + 2_000_000)] = l
    del sim
    files = spool.close() if spool is not None else None
    yield dict(ka=ka, ctx=ctx, index=index, idx_path=idx_path, words=words, lens=lens, n=n, L=L, rec=rec, prefix=prefix, files=files)
    ctx.close()
    if spool is not None:
        spool.remove()


def _run(w, ranges):
    ctx, ka = w["ctx"], w["ka"]
    ctx.reset()
    opts = ka.QuantOpts(1, 0.0, 0.0, 0, 0)
    for a, b in ranges:
        ctx.pseudoalign(opts, w["words"][a * 2 * w["rec"]:b * 2 * w["rec"]], w["lens"][2 * a:2 * b], b - a, w["L"])
    st = ctx.stats()
    return ctx.finalize(), st


def test_full_size_properties(world):
    w = world
    n = w["n"]
    whole, st = _run(w, [(0, n)])
    m_whole = whole.multiset()
    assert st["n_processed"] == n
    total = int(whole.counts.sum(dtype=np.uint64))
    assert 0.99 * n < total <= n                                       # almost every synthetic pair pseudoaligns
    assert len(m_whole) == len(whole.counts)                            # final ECs are distinct sets
    # batching invariance (ragged batch sizes, one of them tiny)
    cuts = [0, 1, 7_000_003, 7_000_260, 19_999_999, n]
    batched, _ = _run(w, list(zip(cuts[:-1], cuts[1:])))
    assert batched.multiset() == m_whole
    # sharding linearity
    a, _ = _run(w, [(0, n // 2)])
    ma = a.multiset()
    b, _ = _run(w, [(n // 2, n)])
    for k, v in b.multiset().items():
        ma[k] = ma.get(k, 0) + v
    assert ma == m_whole
    # EM conservation on the whole problem
    import kallisto_amd.api as A
    ctx, ka, index = w["ctx"], w["ka"], w["index"]
    whole2, _ = _run(w, [(0, n)])
    flens, used = ctx.fld_from_batch(ka.QuantOpts(1, 0.0, 0.0, 0, 0), w["words"], w["lens"], n, w["L"])
    assert used == 10000 and flens.sum() == 10000
    eff = A.eff_lens(index.target_lens, A.mean_frag_lens_trunc(flens))
    alpha, abz, rounds = ctx.em_run(eff)
    assert rounds > 50
    assert abs(alpha.sum() - total) <= 1e-6 * total                     # each EM round redistributes the counts
    tpm = A.counts_to_tpm(alpha, eff)
    assert abs(tpm.sum() - 1e6) < 1e-3
    assert np.all(alpha[abz < 1e-8] == 0)                               # the final clamp (EMAlgorithm.h:217-219)
    balpha, brounds, samp = ctx.bootstrap(12345, eff, want_sample=True)
    assert int(samp.sum(dtype=np.uint64)) == total and abs(balpha.sum() - total) <= 1e-6 * total


def test_prefix_against_oracle(world):
    from oracle import oracle as O
    w = world
    ctx, ka = w["ctx"], w["ka"]
    r1, r2 = w["prefix"][0][:20000], w["prefix"][1][:20000]
    k = r1.shape[0]
    ctx.reset()
    res = ka.quant(ctx, ka.QuantOpts(1, 0.0, 0.0, 0, 0), [(w["words"][:k * 2 * w["rec"]], w["lens"][:2 * k], k, w["L"])])
    oix = O.Index(w["idx_path"])
    buf, off, ln = O.pack_read_matrix(r1, r2)
    ores = O.process_reads(oix, O.Opts(1, 0.0, 0.0, 0, 0), buf, off, ln)
    assert res.ecs.multiset() == ores.multiset()
    assert np.array_equal(res.flens, ores.flens)


def test_prefix_against_reference(world):
    """BASELINE config #3's index, 200 k pairs: the HIP path against the reference binary itself run deterministically."""
    import bench
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "dump_ec")):
        pytest.skip("oracle/_ref/dump_ec not built")
    w = world
    ctx, ka = w["ctx"], w["ka"]
    r1, r2 = w["prefix"]
    k = r1.shape[0]
    ctx.reset()
    res = ka.quant(ctx, ka.QuantOpts(1, 0.0, 0.0, 0, 0), [(w["words"][:k * 2 * w["rec"]], w["lens"][:2 * k], k, w["L"])])
    rep = bench.reference_parity(w["idx_path"], r1, r2, res)
    assert rep["ec_multiset_equal"] and rep["flens_equal"] and rep["eff_length_equal"], rep
    assert rep["est_counts_max_rel_err_tpm_ge_1e-3"] <= 1e-4 and rep["tpm_max_rel_err_tpm_ge_1e-3"] <= 1e-4, rep
    assert rep["tpm_max_abs_err_below_floor"] <= 1e-7 and rep["zero_pattern_equal"], rep
    assert rep["ok"]



def test_whole_run_against_reference(world):
    """BASELINE config #3 at full size: every one of the 30 M pairs through the unmodified reference (dump_ec on all cores; its EM with the
    run's input-order fragment-length sample, src/EMAlgorithm.h:112-223): EC multiset of the WHOLE run, effective lengths and EM round
    count identical, est_counts / TPM within 1e-4; and the oracle's EM on the GPU's own equivalence classes within 1e-9."""
    import bench
    w = world
    if w["files"] is None:
        pytest.skip("oracle/_ref/dump_ec not built or no room for 13 GB of FASTQ")
    ctx, ka, n = w["ctx"], w["ka"], w["n"]
    ctx.reset()
    res = ka.quant(ctx, ka.QuantOpts(1, 0.0, 0.0, 0, 0), [(w["words"], w["lens"], n, w["L"])], download_ecs=True)
    rep = bench.FullSizeParity(w["idx_path"], w["files"], res, min(bench.effective_cpus(), 64)).finish()
    assert rep.get("ec_multiset_equal") and rep["n_processed_ref"] == n, rep
    assert rep["eff_length_equal"] and rep["em_rounds"][0] == rep["em_rounds"][1] and rep["em_rounds"][0] > 50, rep
    assert rep["est_counts_max_rel_err_tpm_ge_1e-3"] <= 1e-4 and rep["tpm_max_rel_err_tpm_ge_1e-3"] <= 1e-4 and rep["zero_pattern_equal"], rep
    assert rep["oracle_em_on_the_gpus_ecs"]["ok"], rep
    assert rep["ok"], rep


def test_tail_against_reference(world):
    """the LAST 200 k pairs of the 30 M, pseudoaligned as the final batch of a run over all of them (record stream recycled, tuple table
    regrown): whole run minus the run without them must be the reference's EC multiset of those pairs at -t 1"""
    import bench
    import torch
    from kallisto_amd.synth_gpu import ReadSimulator
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "dump_ec")):
        pytest.skip("oracle/_ref/dump_ec not built")
    w = world
    ctx, ka, n, L, rec = w["ctx"], w["ka"], w["n"], w["L"], w["rec"]
    # the last chunk's reads again (same generator, same seed, same chunks as the fixture)
    cat, tlens, _ = bench.prepare_workload("human", 20000, True)
    sim = ReadSimulator(cat, tlens, torch.device("cuda", 0), seed=4242, read_len=L)
    for s in range(0, n, 2_000_000):
        r1, r2 = sim.draw(2_000_000)
    k = 200_000
    t1, t2 = r1[-k:].cpu().numpy(), r2[-k:].cpu().numpy()
    del sim, r1, r2
    rep = bench.tail_parity(ctx, ka.QuantOpts(1, 0.0, 0.0, 0, 0), w["idx_path"], w["words"], w["lens"], n, 2, rec, L, t1, t2)
    assert rep["ok"] and rep["counts_never_decrease"] and rep["ec_multiset_equal"], rep
