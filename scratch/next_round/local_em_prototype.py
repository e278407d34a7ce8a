"""Executable spec for the round-2 EM redesign ("component-local EM"): the EC x transcript matrix is block diagonal over the
connected components of the transcript/EC graph, EMAlgorithm::run never couples two components, so a group of components that
fits one workgroup's LDS can iterate on its own -- no kernel boundary or grid barrier per round.  Only the stop rule
(chcount == 0 && i > min_rounds, EMAlgorithm.h:202-205) is global: groups run a chunk of rounds speculatively and record
their per-round change counts, the summed history decides the stopping round, the chunk is replayed up to it from the
checkpoint, then the clamped final round (this is what kamd_em_run_partitioned already does across ranks).

This file checks (numpy, CPU) that the scheme reproduces the oracle's EM exactly in rounds and to 1e-12 in alpha, and
prints the LDS footprint of the groups for a given matrix.  Run: python scratch/next_round/local_em_prototype.py"""
import sys
import numpy as np
import scipy.sparse as sp
import scipy.sparse.csgraph as cg

sys.path.insert(0, ".")
ALPHA_LIMIT, ALPHA_CHANGE_LIMIT, ALPHA_CHANGE, DENORM = 1e-7, 1e-2, 1e-2, 4.9406564584124654e-324


def components(off, ids, T):
    ln = np.diff(off)
    first = np.zeros(len(ln), np.int64)
    nz = ln > 0
    first[nz] = ids[off[:-1][nz]]
    A = sp.coo_matrix((np.ones(len(ids), np.int8), (ids, np.repeat(first, ln))), shape=(T, T)).tocsr()
    nc, lab = cg.connected_components(A, directed=False)
    return nc, lab, first


def groups_by_nnz(off, ids, T, n_groups, min_len=2):
    """components (of the rows with >= min_len members) in hashed order, cut where the running nnz passes k * total / n_groups"""
    nc, lab, first = components(off, ids, T)
    ln = np.diff(off)
    keep = ln >= min_len
    nnz_c = np.bincount(lab[first[keep]], weights=ln[keep].astype(float), minlength=nc)
    order = np.argsort((np.arange(nc, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(20), kind="stable")
    cum = np.cumsum(nnz_c[order])
    grp_of_comp = np.empty(nc, np.int64)
    grp_of_comp[order] = np.minimum((cum - 1e-9) * n_groups // max(cum[-1], 1), n_groups - 1).astype(np.int64)
    return lab, first, grp_of_comp, nnz_c


def footprint(off, ids, T, n_groups):
    lab, first, goc, nnz_c = groups_by_nnz(off, ids, T, n_groups)
    ln = np.diff(off)
    keep = ln >= 2
    g_row = goc[lab[first]]
    nnz_g = np.bincount(g_row[keep], weights=ln[keep].astype(float), minlength=n_groups)
    rows_g = np.bincount(g_row[keep], minlength=n_groups)
    in_multi = np.zeros(T, bool)
    in_multi[ids[np.repeat(keep, ln)]] = True
    tr_g = np.bincount(goc[lab[np.nonzero(in_multi)[0]]], minlength=n_groups)
    # LDS bytes: 2 x u16 index streams, row/col offsets (u32), g + cnt/wc per row, a/alpha/single/eff per transcript
    lds = nnz_g * 4 + rows_g * (4 + 8 + 8) + tr_g * (4 + 8 * 4)
    return dict(groups=n_groups, largest_component_nnz=int(nnz_c.max()), max_nnz=int(nnz_g.max()), max_rows=int(rows_g.max()),
                max_tr=int(tr_g.max()), max_lds_kb=round(float(lds.max()) / 1024, 1))


def em_local(off, ids, counts, eff, T, n_groups=8, chunk=64, n_iter=10000, min_rounds=50):
    """the scheme; every group iterates independently inside a chunk"""
    lab, first, goc, _ = groups_by_nnz(off, ids, T, n_groups, min_len=1)
    ln = np.diff(off)
    g_row = goc[lab[first]]
    rows = np.repeat(np.arange(len(ln)), ln)
    wc = counts                                            # weight counts = counts (main run)
    alpha = np.full(T, 1.0 / T)

    def run_rounds(alpha0, n, final, hist):
        out = alpha0.copy()
        for g in range(n_groups):                          # one workgroup each; no communication
            rsel = np.nonzero(g_row == g)[0]
            if len(rsel) == 0:
                continue
            esel = np.nonzero(np.isin(rows, rsel))[0]
            tl = np.unique(ids[esel])
            a = out[tl].copy()
            loc = {t: i for i, t in enumerate(tl)}
            e_t = np.array([loc[t] for t in ids[esel]])
            e_r = np.searchsorted(rsel, rows[esel])
            for r in range(n):
                cur = a.copy()
                if final:
                    cur[cur < ALPHA_LIMIT / 10] = 0.0
                w = cur[e_t] * (wc[rsel][e_r] / eff[tl][e_t])        # weight_map * alpha (EMAlgorithm.h:140-150)
                denom = np.bincount(e_r, weights=w, minlength=len(rsel))
                ok = (counts[rsel] > 0) & (denom >= DENORM)
                cn = np.where(ok, counts[rsel] / np.where(ok, denom, 1.0), 0.0)
                nxt = np.bincount(e_t, weights=w * cn[e_r], minlength=len(tl))
                ch = int(np.sum((nxt > ALPHA_CHANGE_LIMIT) & (np.abs(nxt - cur) / np.where(nxt > 0, nxt, 1.0) > ALPHA_CHANGE)))
                if hist is not None:
                    hist[r] += ch
                a = nxt
            out[tl] = a
        # transcripts in no row keep alpha -> next = 0 after the first round (the reference zeroes next_alpha every round)
        untouched = np.ones(T, bool)
        untouched[ids] = False
        if n > 0:
            out[untouched] = 0.0
        return out

    base = 0
    while True:
        n = min(chunk, n_iter - base)
        hist = np.zeros(n, np.int64)
        nxt = run_rounds(alpha, n, False, hist)
        stop = next((base + i for i in range(n) if hist[i] == 0 and base + i > min_rounds), None)
        if stop is None:
            alpha, base = nxt, base + n
            if base >= n_iter:
                return alpha, None, n_iter
            continue
        alpha = run_rounds(alpha, stop - base + 1, False, None)     # replay from the checkpoint
        abz = alpha.copy()
        alpha = run_rounds(alpha, 1, True, None)                    # the clamped final round
        return alpha, abz, stop + 1


def gene_matrix(n_genes, seed):
    """gene families only (no hubs): one component per gene"""
    rng = np.random.default_rng(seed)
    iso = np.minimum(rng.geometric(0.12, n_genes), 40)
    t0 = np.concatenate([[0], np.cumsum(iso)])
    T = int(t0[-1]) + 3
    sets = {}
    for g in range(n_genes):
        for _ in range(rng.poisson(20)):
            m = min(iso[g], max(1, rng.geometric(0.25)))
            r = tuple(sorted(int(x) + int(t0[g]) for x in rng.choice(iso[g], m, replace=False)))
            sets[r] = sets.get(r, 0) + int(rng.pareto(1.2) * 3) + (1 if rng.random() < 0.9 else 0)
    keys = list(sets)
    rng.shuffle(keys)
    off = np.zeros(len(keys) + 1, np.uint64)
    off[1:] = np.cumsum([len(k) for k in keys])
    return off, np.array([t for k in keys for t in k], np.uint32), np.array([sets[k] for k in keys], np.uint32), rng.uniform(150, 3000, T), T


if __name__ == "__main__":
    from oracle import oracle as O
    from tests.test_gpu_parity import _family_csr
    for name, (off, ids, cnt, eff, T) in (("genes only", gene_matrix(150, 5)), ("hubs + long rows (one giant component)", _family_csr(400, 7))):
        a_o, abz_o, r_o = O.em_run(off, ids, cnt, eff, T)
        for ng in (1, 7, 32):
            a, abz, r = em_local(off.astype(np.int64), ids.astype(np.int64), cnt.astype(np.float64), eff, T, n_groups=ng)
            rel = np.max(np.abs(a - a_o) / np.maximum(np.abs(a_o), 1e-9))
            print(f"{name}: groups {ng:3d}: rounds {r} (oracle {r_o}), max rel diff of alpha {rel:.2e}")
            assert r == r_o and rel < 1e-9
        print("   footprint:", footprint(off.astype(np.int64), ids.astype(np.int64), T, 16))
