"""Synthetic transcriptomes and reads for the BASELINE.json workloads.

No real transcriptome is available offline, so every workload is generated from fixed seeds:

* ``yeast_like``  -- config #2: ~6 k transcripts, log-normal lengths, 5 % paralog-like shared segments.
* ``human_like``  -- configs #3-5: genes x isoforms built by exon skipping / alternative ends from a per-gene
  exon pool, so that the transcriptome de Bruijn graph has realistic mosaic equivalence classes.

The transcriptome is written as FASTA and indexed by the *reference* binary (``kallisto index``), so that the
reference and this implementation share one index file.  Reads are drawn as fragments of transcripts with a
log-normal expression profile, substitution errors and occasional ``N``.

This module is workload generation only (numpy); nothing here is on the product path.
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def yeast_like(n_tr: int = 6000, seed: int = 1):
    """Return list of uint8 ASCII arrays (one per transcript)."""
    rng = np.random.default_rng(seed)
    lens = np.clip(np.exp(rng.normal(np.log(1300.0), 0.6, n_tr)), 300, 8000).astype(np.int64)
    seqs = [_ACGT[rng.integers(0, 4, int(l))] for l in lens]
    # 5 % of transcripts share a 200-600 bp segment with another one (paralog-like)
    n_par = n_tr // 20
    for t in rng.choice(n_tr, n_par, replace=False):
        o = int(rng.integers(0, n_tr))
        if o == t:
            continue
        seg = int(rng.integers(200, 601))
        seg = min(seg, len(seqs[o]), len(seqs[t]))
        so = int(rng.integers(0, len(seqs[o]) - seg + 1))
        st = int(rng.integers(0, len(seqs[t]) - seg + 1))
        seqs[t] = seqs[t].copy()
        seqs[t][st:st + seg] = seqs[o][so:so + seg]
    return seqs


def human_like(n_genes: int = 20000, mean_iso: float = 10.0, seed: int = 2):
    """Genes with 4-20 exons of 50-400 bp; isoforms = exon-skipping + alternative first/last exon ends."""
    rng = np.random.default_rng(seed)
    seqs = []
    for g in range(n_genes):
        n_ex = int(rng.integers(4, 21))
        ex_len = rng.integers(50, 401, n_ex)
        exons = [_ACGT[rng.integers(0, 4, int(l))] for l in ex_len]
        n_iso = int(min(rng.geometric(1.0 / mean_iso), 60))
        seen = set()
        for _ in range(n_iso):
            keep = rng.random(n_ex) < 0.75
            if keep.sum() < 2:
                keep[:2] = True
            idx = np.flatnonzero(keep)
            # alternative ends: trim the first / last exon of the isoform with prob 0.3
            a = int(rng.integers(0, ex_len[idx[0]] - 40)) if rng.random() < 0.3 else 0
            b = int(rng.integers(0, ex_len[idx[-1]] - 40)) if rng.random() < 0.3 else 0
            key = (idx.tobytes(), a, b)
            if key in seen:
                continue
            seen.add(key)
            parts = [exons[i] for i in idx]
            parts[0] = parts[0][a:]
            if b:
                parts[-1] = parts[-1][:-b]
            seqs.append(np.concatenate(parts))
    return seqs


def write_fasta(path: str, seqs, prefix: str = "tr") -> None:
    with open(path, "wb") as f:
        for i, s in enumerate(seqs):
            f.write(b">%s%d\n" % (prefix.encode(), i))
            f.write(s.tobytes())
            f.write(b"\n")


def revcomp(a: np.ndarray) -> np.ndarray:
    return _COMP[a[..., ::-1]]


def simulate_reads(seqs, n: int, read_len: int = 100, paired: bool = True, frag_mean: float = 200.0,
                   frag_sd: float = 30.0, err: float = 0.005, n_frac: float = 0.001, seed: int = 3,
                   expr_sigma: float = 2.0):
    """Return (r1, r2) uint8 ASCII arrays of shape (n, read_len) (r2 is None for single-end).

    Fragments are drawn from transcripts with probability ~ expression x length; fragment length is
    N(frag_mean, frag_sd) truncated to [read_len, 999]; half of the fragments come from the reverse strand;
    mate 2 is the reverse complement of the fragment's 3' end.
    """
    rng = np.random.default_rng(seed)
    T = len(seqs)
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    off = np.zeros(T + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    cat = np.concatenate(seqs)
    expr = np.exp(rng.normal(0.0, expr_sigma, T))
    fl = np.clip(np.rint(rng.normal(frag_mean, frag_sd, n)), read_len, 999).astype(np.int64)
    w = expr * np.maximum(lens - frag_mean, 1)
    w[lens < read_len] = 0
    tr = rng.choice(T, n, p=w / w.sum())
    fl = np.minimum(fl, lens[tr])
    start = (rng.random(n) * (lens[tr] - fl + 1)).astype(np.int64)
    base = off[tr] + start
    ar = np.arange(read_len, dtype=np.int64)
    left = cat[base[:, None] + ar[None, :]]                      # 5' end of fragment, forward strand
    right = cat[(base + fl - read_len)[:, None] + ar[None, :]]   # 3' end of fragment, forward strand
    flip = rng.random(n) < 0.5
    r1 = np.where(flip[:, None], revcomp(right), left)
    r2 = np.where(flip[:, None], left, revcomp(right)) if paired else None
    # errors (r1 and r2 are already mate-oriented; mate2 = revcomp of the other fragment end)
    for r in (r1, r2):
        if r is None:
            continue
        m = rng.random(r.shape) < err
        r[m] = _ACGT[rng.integers(0, 4, int(m.sum()))]
        nm = rng.random(n) < n_frac
        pos = rng.integers(0, read_len, n)
        r[np.flatnonzero(nm), pos[nm]] = ord("N")
    return r1, r2


def write_fastq(path: str, reads: np.ndarray, tag: str = "r") -> None:
    n, L = reads.shape
    qual = b"I" * L
    with open(path, "wb") as f:
        chunk = 100000
        for s in range(0, n, chunk):
            out = []
            for i in range(s, min(n, s + chunk)):
                out.append(b"@%s%d\n%s\n+\n%s\n" % (tag.encode(), i, reads[i].tobytes(), qual))
            f.write(b"".join(out))
