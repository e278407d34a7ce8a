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


def _mutate(rng, seq: np.ndarray, div: float, indel: float = 0.0) -> np.ndarray:
    """a copy of seq with a fraction `div` of substituted bases (always to another base) and, with `indel` > 0, that fraction of single-base deletions"""
    out = seq.copy()
    m = rng.random(out.size) < div
    if m.any():
        cur = np.searchsorted(_ACGT, out[m])           # A C G T are sorted in ASCII
        out[m] = _ACGT[(cur + rng.integers(1, 4, int(m.sum()))) & 3]
    if indel > 0.0:
        keep = rng.random(out.size) >= indel
        out = out[keep]
    return out


def human_stress(n_genes: int = 20000, mean_iso: float = 10.0, seed: int = 7, n_repeat_families: int = 4, repeat_len: int = 300,
                 repeat_gene_frac: float = 0.3, repeat_max_div: float = 0.15, n_paralog_families: int = 3, paralog_members: int = 200,
                 polya_frac: float = 0.03, background_mbp: float = 8.0):
    """`human_like` with the structure a real transcriptome has and a uniform random one lacks (VERDICT r4 "What's missing" #1):

    * **repeat families** (Alu / L1-like): `n_repeat_families` consensus elements of about `repeat_len` bp; a copy -- 0 .. `repeat_max_div`
      diverged from the consensus (skewed to young, near-identical copies), sometimes truncated, in either orientation -- sits in the 3' or 5'
      terminal exon ("UTR") of `repeat_gene_frac` of the genes, i.e. in every isoform that keeps that exon.  Reads from a copy fall into
      equivalence classes that span unrelated genes: one such family chains thousands of genes into ONE connected component of the
      transcript/EC graph;
    * **paralog families**: `n_paralog_families` x `paralog_members` genes that are 1-10 % diverged copies of one ancestral gene (classes of
      hundreds of transcripts);
    * **poly-A tails** of 31-60 A on `polya_frac` of the transcripts: the all-A k-mer is shared by thousands of transcripts (the classes
      `--ec-max-size` exists for, /root/reference/src/main.cpp:2151);
    * a **background** ("genome": random sequence with copies of the same repeat families sprinkled in at ~10 % density) from which the read
      simulators draw off-transcriptome ("intronic" / intergenic) fragments -- reads that share k-mers with the UTR copies without coming from
      a transcript.

    Returns (seqs, background): a list of uint8 ASCII arrays (one per transcript) and one uint8 ASCII array."""
    rng = np.random.default_rng(seed)
    fam_len = rng.integers(int(repeat_len * 0.8), int(repeat_len * 1.2) + 1, n_repeat_families)
    fams = [_ACGT[rng.integers(0, 4, int(l))] for l in fam_len]

    def repeat_copy():
        f = int(rng.integers(0, n_repeat_families))
        div = repeat_max_div * float(rng.random()) ** 2          # half of the copies are < 4 % diverged
        c = _mutate(rng, fams[f], div, indel=div / 10.0)
        if rng.random() < 0.3 and c.size > 200:                    # truncated copies
            a = int(rng.integers(0, c.size - 150))
            c = c[a:a + int(rng.integers(150, c.size - a + 1))]
        return revcomp(c) if rng.random() < 0.5 else c

    # paralog families: the ancestral exon pools
    par_genes = {}
    paralog_members = min(paralog_members, n_genes // max(4 * n_paralog_families, 1))   # (small test transcriptomes)
    if n_paralog_families and paralog_members:
        chosen = rng.choice(n_genes, n_paralog_families * paralog_members, replace=False)
        for f in range(n_paralog_families):
            n_ex = int(rng.integers(6, 15))
            anc = [_ACGT[rng.integers(0, 4, int(l))] for l in rng.integers(80, 401, n_ex)]
            for g in chosen[f * paralog_members:(f + 1) * paralog_members]:
                par_genes[int(g)] = anc
    seqs = []
    for g in range(n_genes):
        if g in par_genes:
            d = float(rng.uniform(0.01, 0.10))
            exons = [_mutate(rng, e, d) for e in par_genes[g]]
            n_ex = len(exons)
        else:
            n_ex = int(rng.integers(4, 21))
            exons = [_ACGT[rng.integers(0, 4, int(l))] for l in rng.integers(50, 401, n_ex)]
        if rng.random() < repeat_gene_frac:                         # a repeat copy inside a terminal exon
            k = n_ex - 1 if rng.random() < 0.75 else 0
            at = int(rng.integers(0, exons[k].size + 1))
            exons[k] = np.concatenate([exons[k][:at], repeat_copy(), exons[k][at:]])
        ex_len = np.array([e.size for e in exons])
        n_iso = int(min(rng.geometric(1.0 / mean_iso), 60))
        seen = set()
        for _ in range(n_iso):
            keep = rng.random(n_ex) < 0.75
            if keep.sum() < 2:
                keep[:2] = True
            idx = np.flatnonzero(keep)
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
            if rng.random() < polya_frac:
                parts.append(np.full(int(rng.integers(31, 61)), ord("A"), np.uint8))
            seqs.append(np.concatenate(parts))
    # the background: random sequence, ~10 % of it copies of the repeat families
    n_bg = int(background_mbp * 1e6)
    bg = _ACGT[rng.integers(0, 4, n_bg)]
    n_copies = int(0.10 * n_bg / max(repeat_len, 1))
    for at in rng.integers(0, max(n_bg - 2 * repeat_len, 1), n_copies):
        c = repeat_copy()
        bg[at:at + c.size] = c[:max(0, min(c.size, n_bg - int(at)))]
    return seqs, bg


def write_fasta(path: str, seqs, prefix: str = "tr") -> None:
    with open(path, "wb") as f:
        for i, s in enumerate(seqs):
            f.write(b">%s%d\n" % (prefix.encode(), i))
            f.write(s.tobytes())
            f.write(b"\n")


def revcomp(a: np.ndarray) -> np.ndarray:
    return _COMP[a[..., ::-1]]


def tail_error_profile(read_len: int, err: float, tail_err: float, mate: int = 0) -> np.ndarray:
    """per-position substitution probability: `err` everywhere plus a 3' quality tail that rises with the fourth power of the position to
    `tail_err` at the last base (mate 2 degrades 1.5 x as fast) -- the shape of an Illumina run's error profile"""
    p = np.arange(read_len, dtype=np.float64) / max(read_len - 1, 1)
    return np.minimum(err + tail_err * (1.5 if mate else 1.0) * p ** 4, 0.5)


# the four quality bins of a NovaSeq-style run (Phred 2, 12, 23, 37)
_QBINS = np.frombuffer(b"#-8F", dtype=np.uint8)


def quality_strings(rng, err_prob: np.ndarray, is_error: np.ndarray) -> np.ndarray:
    """binned Phred characters for an (n, L) batch: the bin of the position's error probability, demoted where the base actually is wrong (70 %)
    and at a random 3 % of the positions"""
    q = -10.0 * np.log10(np.maximum(err_prob, 1e-5))
    base = np.where(q >= 30, 3, np.where(q >= 20, 2, np.where(q >= 10, 1, 0))).astype(np.int64)   # (L,)
    b = np.broadcast_to(base[None, :], is_error.shape).copy()
    demote = (is_error & (rng.random(is_error.shape) < 0.7)) | (rng.random(is_error.shape) < 0.03)
    b[demote] = np.maximum(b[demote] - rng.integers(1, 3, int(demote.sum())), 0)
    return _QBINS[b]


def simulate_reads(seqs, n: int, read_len: int = 100, paired: bool = True, frag_mean: float = 200.0,
                   frag_sd: float = 30.0, err: float = 0.005, n_frac: float = 0.001, seed: int = 3,
                   expr_sigma: float = 2.0, background=None, off_frac: float = 0.0, tail_err: float = 0.0,
                   return_qual: bool = False):
    """Return (r1, r2) uint8 ASCII arrays of shape (n, read_len) (r2 is None for single-end); with return_qual (r1, r2, q1, q2).

    Fragments are drawn from transcripts with probability ~ expression x length; fragment length is
    N(frag_mean, frag_sd) truncated to [read_len, 999]; half of the fragments come from the reverse strand;
    mate 2 is the reverse complement of the fragment's 3' end.
    Stress options (human_stress): a fraction `off_frac` of the fragments comes from `background` instead of a transcript (half of those from
    uniform random sequence when the background is shorter than a fragment); `tail_err` adds the 3' quality tail of tail_error_profile.
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
    if background is not None and off_frac > 0.0:
        off = np.flatnonzero(rng.random(n) < off_frac)
        bstart = (rng.random(off.size) * (background.size - 1000)).astype(np.int64)
        left[off] = background[bstart[:, None] + ar[None, :]]
        right[off] = background[(bstart + fl[off] - read_len)[:, None] + ar[None, :]]
    flip = rng.random(n) < 0.5
    r1 = np.where(flip[:, None], revcomp(right), left)
    r2 = np.where(flip[:, None], left, revcomp(right)) if paired else None
    # errors (r1 and r2 are already mate-oriented; mate2 = revcomp of the other fragment end)
    quals = []
    for mate, r in enumerate((r1, r2)):
        if r is None:
            quals.append(None)
            continue
        prof = tail_error_profile(read_len, err, tail_err, mate) if tail_err > 0.0 else np.full(read_len, err)
        m = rng.random(r.shape) < prof[None, :]
        r[m] = _ACGT[rng.integers(0, 4, int(m.sum()))]
        nm = rng.random(n) < n_frac
        pos = rng.integers(0, read_len, n)
        r[np.flatnonzero(nm), pos[nm]] = ord("N")
        if return_qual:
            q = quality_strings(rng, prof, m)
            q[np.flatnonzero(nm), pos[nm]] = ord("#")
            quals.append(q)
    if return_qual:
        return r1, r2, quals[0], quals[1]
    return r1, r2


def illumina_headers(n: int, mate: int, seed: int = 11, start: int = 0):
    """n Illumina-style read names of VARIABLE length (instrument:run:flowcell:lane:tile:x:y mate:N:0:index), the same for both mates up to the
    mate digit: what a sequencer writes, unlike the fixed 11-byte `@r000000001` of write_fastq_fast"""
    rng = np.random.default_rng(seed)
    rng2 = np.random.default_rng(seed + 1 + start)
    lane = rng.integers(1, 5)
    tile = 1101 + (np.arange(start, start + n) // 40000) % 78 + 1000 * ((np.arange(start, start + n) // 3120000) % 2)
    x = rng2.integers(1000, 32768, n)
    y = rng2.integers(1000, 65536, n)
    return [b"@A00587:214:H7FKL2DSXY:%d:%d:%d:%d %d:N:0:ACGTTGCA+TTGACCAA" % (lane, tile[i], x[i], y[i], mate + 1) for i in range(n)]


def write_fastq_realistic(path: str, reads: np.ndarray, quals: np.ndarray, mate: int = 0, seed: int = 11) -> None:
    """FASTQ as a sequencer writes it: variable-length Illumina headers, a real quality string per read"""
    n = reads.shape[0]
    with open(path, "wb") as f:
        chunk = 100000
        for s in range(0, n, chunk):
            m = min(chunk, n - s)
            hd = illumina_headers(m, mate, seed, s)
            f.write(b"".join(b"%s\n%s\n+\n%s\n" % (hd[i], reads[s + i].tobytes(), quals[s + i].tobytes()) for i in range(m)))


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
