"""FASTQ text as a sequencer writes it, assembled with torch tensor ops on whatever device the reads live on (bench.py's FastqSpool
on the GPU, its small samples on the CPU).  Workload generation only; nothing here is product code.

A record is `@<instrument>:<run>:<flowcell>:<lane>:<tile>:<x>:<y> <mate>:N:0:<index>`, the read, `+`, a quality string:
  * the header has a VARIABLE length (x and y have four or five digits; 63-65 bytes in all instead of the 11 of `@r000000001`), the same for
    both mates up to the mate digit -- what kseq's record grammar (src/kseq.h:174-215) and every FASTQ front-end has to walk through;
  * the quality string is the binned Phred profile of a NovaSeq-style run (bins F : , #) falling off towards the 3' end, with a few per cent of
    demoted positions -- not 100 x `I`: half of a real deflate stream is quality bytes, and a constant line turns them into one long match.
The sequences are whatever the caller's simulator produced; the qualities are cosmetic (kallisto ignores them).
"""
from __future__ import annotations

import numpy as np
import torch

PREFIX = b"@A00587:214:H7FKL2DSXY:"
SUFFIX = b":N:0:ACGTTGCA+TTGACCAA\n"
QBINS = b"#,:F"


def _digits(v: torch.Tensor, width: int) -> torch.Tensor:
    p = (10 ** torch.arange(width - 1, -1, -1, device=v.device, dtype=torch.int64))[None, :]
    return ((v[:, None] // p) % 10 + 48).to(torch.uint8)


def quality_profile(L: int, mate: int) -> np.ndarray:
    """bin index (0..3) per position: Phred 37 over most of the read, 25 / 11 towards the 3' end (mate 2 earlier)"""
    p = np.arange(L) / max(L - 1, 1)
    e = 0.0003 + 0.03 * (1.5 if mate else 1.0) * p ** 8
    q = -10.0 * np.log10(e)
    return np.where(q >= 30, 3, np.where(q >= 20, 2, np.where(q >= 10, 1, 0))).astype(np.int64)


class FastqText:
    """records of one mate file: text(reads, first_index) -> 1-D uint8 tensor (the concatenated records of the chunk)"""

    def __init__(self, L: int, mate: int, device, seed: int = 77):
        self.L, self.mate, self.dev = L, mate, device
        self.h0 = len(PREFIX)
        # fixed-width layout of a record with five-digit coordinates; the leading digit of a four-digit coordinate is dropped when the text is cut
        #   PREFIX lane : tile(4) : x(5) : y(5) ' ' mate SUFFIX seq \n + \n qual \n
        o = self.h0
        self.o_lane = o; o += 2
        self.o_tile = o; o += 5
        self.o_x = o; o += 6
        self.o_y = o; o += 5
        self.o_sp = o; o += 1
        self.o_mate = o; o += 1
        self.o_suf = o; o += len(SUFFIX)
        self.o_seq = o; o += L + 1
        self.o_plus = o; o += 2
        self.o_q = o; o += L + 1
        self.width = o
        self.base = torch.from_numpy(quality_profile(L, mate)).to(device)
        self.qb = torch.tensor(list(QBINS), dtype=torch.uint8, device=device)
        self.pos = torch.arange(L, device=device, dtype=torch.int64)[None, :]
        self.seed = seed
        self._tmpl = None

    def _template(self, m):
        if self._tmpl is None or self._tmpl.shape[0] < m:
            t = torch.zeros((m, self.width), dtype=torch.uint8, device=self.dev)
            t[:, :self.h0] = torch.tensor(list(PREFIX), dtype=torch.uint8, device=self.dev)
            t[:, self.o_lane + 1] = ord(":"); t[:, self.o_tile + 4] = ord(":"); t[:, self.o_x + 5] = ord(":"); t[:, self.o_sp] = ord(" ")
            t[:, self.o_mate] = ord("1") + self.mate
            t[:, self.o_suf:self.o_suf + len(SUFFIX)] = torch.tensor(list(SUFFIX), dtype=torch.uint8, device=self.dev)
            t[:, self.o_seq + self.L] = 10; t[:, self.o_plus] = ord("+"); t[:, self.o_plus + 1] = 10; t[:, self.o_q + self.L] = 10
            self._tmpl = t
        return self._tmpl[:m]

    def text(self, reads: torch.Tensor, first: int) -> torch.Tensor:
        m, L = reads.shape
        assert L == self.L
        t = self._template(m)
        idx = torch.arange(first, first + m, device=self.dev, dtype=torch.int64)
        # coordinates: a fixed scramble of the record index (the same for both mates), 1000 .. 32767 / 1000 .. 65535
        hx = (idx * 2654435761 + 12345) & 0x7FFFFFFF
        hy = (idx * 40503 + 977) * 2246822519 & 0x7FFFFFFF
        x = 1000 + hx % 31768
        y = 1000 + hy % 64536
        t[:, self.o_lane] = (49 + (idx // 3_120_000) % 4).to(torch.uint8)
        t[:, self.o_tile:self.o_tile + 4] = _digits(1101 + (idx // 40_000) % 78, 4)
        t[:, self.o_x:self.o_x + 5] = _digits(x, 5)
        t[:, self.o_y:self.o_y + 5] = _digits(y, 5)
        t[:, self.o_seq:self.o_seq + L] = reads
        # qualities: the profile's bin, 3 % of the positions demoted by one or two bins, 'N' -> the lowest bin
        b = self.base[None, :].expand(m, L).clone()
        # (a fixed scramble of (record, position, mate): the text of a record does not depend on how the reads were cut into chunks)
        hq = ((idx[:, None] * 1000003 + self.pos * 7919 + self.mate * 131 + self.seed) * 2654435761) & 0x7FFFFFFF
        r = (hq % 10007).double() / 10007.0
        b = torch.where(r < 0.03, (b - 1 - (r < 0.01).long()).clamp_(min=0), b)
        b = torch.where(reads == ord("N"), torch.zeros_like(b), b)
        t[:, self.o_q:self.o_q + L] = self.qb[b]
        keep = torch.ones((m, self.width), dtype=torch.bool, device=self.dev)
        keep[:, self.o_x] = x >= 10000
        keep[:, self.o_y] = y >= 10000
        return t[keep]


def write_fastq(path: str, reads: np.ndarray, mate: int = 0, first: int = 0, chunk: int = 1_000_000, device=None) -> None:
    """numpy (n, L) reads -> a FASTQ file of realistic records; the text is assembled on `device` (the GPU when there is one: the CPU needs
    about 6 us per record)"""
    n, L = reads.shape
    dev = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    ft = FastqText(L, mate, dev)
    with open(path, "wb") as f:
        for s in range(0, n, chunk):
            r = torch.from_numpy(np.ascontiguousarray(reads[s:s + chunk])).to(dev)
            f.write(memoryview(ft.text(r, first + s).cpu().numpy()).cast("B"))
