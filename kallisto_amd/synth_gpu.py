"""Device-side read simulation for bench.py (workload generation only; torch tensor ops, nothing here is product code).

Same model as kallisto_amd.synth.simulate_reads: fragments of N(frag_mean, frag_sd) truncated to [read_len, 999] drawn
from transcripts with probability ~ expression x length, half from the reverse strand, mate 2 = reverse complement of
the fragment's other end, substitution errors, occasional N.  Stress options (synth.human_stress): a fraction of the
fragments from a background sequence instead of a transcript, a 3' quality tail of substitution errors.
"""
from __future__ import annotations

import numpy as np
import torch


class ReadSimulator:
    def __init__(self, cat: np.ndarray, lens: np.ndarray, device, seed: int = 3, expr_sigma: float = 2.0,
                 read_len: int = 100, frag_mean: float = 200.0, frag_sd: float = 30.0, err: float = 0.005,
                 n_frac: float = 0.001, background: np.ndarray | None = None, off_frac: float = 0.0, tail_err: float = 0.0):
        self.dev = device
        self.read_len, self.frag_mean, self.frag_sd, self.err, self.n_frac = read_len, frag_mean, frag_sd, err, n_frac
        self.off_frac, self.tail_err = (off_frac if background is not None else 0.0), tail_err
        self.bg = torch.from_numpy(background).to(device) if background is not None else None
        from .synth import tail_error_profile
        # per-position substitution probability of mate 1 / mate 2 (a constant without the quality tail)
        self.prof = [torch.from_numpy(tail_error_profile(read_len, err, tail_err, m) if tail_err > 0.0 else np.full(read_len, err)).float().to(device)
                     for m in (0, 1)]
        self.cat = torch.from_numpy(cat).to(device)
        lens_t = torch.from_numpy(lens.astype(np.int64)).to(device)
        self.lens = lens_t
        self.off = torch.zeros(len(lens) + 1, dtype=torch.int64, device=device)
        self.off[1:] = torch.cumsum(lens_t, 0)
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        self.g = g
        expr = torch.exp(torch.randn(len(lens), generator=g, device=device, dtype=torch.float64) * expr_sigma)
        w = expr * torch.clamp(lens_t.double() - frag_mean, min=1.0)
        w[lens_t < max(read_len, 120)] = 0
        self.cdf = torch.cumsum(w / w.sum(), 0)
        comp = torch.zeros(256, dtype=torch.uint8, device=device)
        for a, b in zip(b"ACGTN", b"TGCAN"):
            comp[a] = b
        self.comp = comp
        self.acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)

    def draw(self, n: int):
        """-> (r1, r2) uint8 ASCII tensors of shape (n, read_len) on the device."""
        g, dev, L = self.g, self.dev, self.read_len
        u = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
        tr = torch.searchsorted(self.cdf, u).clamp_(max=len(self.lens) - 1)
        fl = torch.round(torch.randn(n, generator=g, device=dev) * self.frag_sd + self.frag_mean).long().clamp_(L, 999)
        fl = torch.minimum(fl, self.lens[tr])
        start = (torch.rand(n, generator=g, device=dev, dtype=torch.float64) * (self.lens[tr] - fl + 1).double()).long()
        base = self.off[tr] + start
        ar = torch.arange(L, device=dev)
        left = self.cat[base[:, None] + ar[None, :]]
        right = self.cat[(base + fl - L)[:, None] + ar[None, :]]
        if self.off_frac > 0.0:   # off-transcriptome fragments ("intronic" / intergenic): the same geometry, drawn from the background
            off = torch.nonzero(torch.rand(n, generator=g, device=dev) < self.off_frac).squeeze(1)
            bstart = (torch.rand(off.numel(), generator=g, device=dev, dtype=torch.float64) * (self.bg.numel() - 1000)).long()
            left[off] = self.bg[bstart[:, None] + ar[None, :]]
            right[off] = self.bg[(bstart + fl[off] - L)[:, None] + ar[None, :]]
        rc_right = self.comp[right.long()].flip(1)
        flip = torch.rand(n, generator=g, device=dev) < 0.5
        # forward fragment: mate 1 = left end, mate 2 = revcomp(right end); reverse-strand fragment: the roles swap
        r1 = torch.where(flip[:, None], rc_right, left)
        r2 = torch.where(flip[:, None], left, rc_right)
        for mate, r in enumerate((r1, r2)):
            m = torch.rand(r.shape, generator=g, device=dev) < self.prof[mate][None, :]
            rnd = self.acgt[torch.randint(0, 4, r.shape, generator=g, device=dev)]
            r[m] = rnd[m]
            nm = torch.rand(n, generator=g, device=dev) < self.n_frac
            pos = torch.randint(0, L, (n,), generator=g, device=dev)
            rows = torch.nonzero(nm).squeeze(1)
            r[rows, pos[rows]] = ord("N")
        return r1.contiguous(), r2.contiguous()
