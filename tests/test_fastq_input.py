"""The front-end's sequence readers (kallisto_amd/csrc/kamd_fastq.h: what FastqSequenceReader::fetchSequences does in
the reference, src/ProcessReads.cpp:3128-3267) on the CPU: plain 4-line FASTQ through the mmap + multi-thread indexer,
gzip and FASTA through the serial reader, BGZF through the block-parallel inflater -- all must yield the sequences a
straightforward Python parse yields, in order."""
import ctypes as C
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from tests import emu_binding as E


def _reads(n, seed, lo=20, hi=151, n_frac=0.02):
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    out = []
    for _ in range(n):
        s = acgt[rng.integers(0, 4, int(rng.integers(lo, hi)))].copy()
        if rng.random() < n_frac:
            s[rng.integers(0, len(s))] = ord("N")
        out.append(bytes(s))
    return out


def _fastq_bytes(reads, crlf=False, tricky_quals=True):
    nl = b"\r\n" if crlf else b"\n"
    parts = []
    for i, r in enumerate(reads):
        q = bytearray(b"I" * len(r))
        if tricky_quals and len(q) and i % 7 == 0:
            q[0] = ord("@")                     # a quality line may start with '@' (and look like a header)
        if tricky_quals and len(q) and i % 11 == 0:
            q[0] = ord("+")
        parts.append(b"@r%d some comment" % i + nl + r + nl + b"+" + nl + bytes(q) + nl)
    return b"".join(parts)


def _bgzf(data, block=65280, level=6):
    """BGZF as `bgzip` writes it: members of <= 64 KiB with the BC extra subfield, closed by the empty EOF member."""
    out = []
    for a in list(range(0, len(data), block)) + [None]:
        chunk = b"" if a is None else data[a:a + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(comp) + 8
        out.append(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
                   + comp + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    return b"".join(out)


def _chunks(path, threads, chunk=1000, cap=1 << 26):
    L = E.lib()
    L.io_read_chunks.restype = C.c_int64
    buf = C.create_string_buffer(cap)
    n = C.c_uint64(0)
    r = L.io_read_chunks(path.encode(), int(threads), C.c_uint64(chunk), buf, C.c_uint64(cap), C.byref(n))
    assert r >= 0, r
    return buf.raw[:r].split(b"\n")[:-1], n.value


def _mapped(path, threads, cap=1 << 26):
    L = E.lib()
    L.io_index_fastq.restype = C.c_int64
    buf = C.create_string_buffer(cap)
    n = C.c_uint64(0)
    r = L.io_index_fastq(path.encode(), int(threads), buf, C.c_uint64(cap), C.byref(n))
    if r == -2:
        return None, 0
    assert r >= 0, r
    return buf.raw[:r].split(b"\n")[:-1], n.value


@pytest.mark.parametrize("threads", [1, 3, 8])
@pytest.mark.parametrize("crlf", [False, True])
def test_plain_fastq_parallel_indexer(tmp_path, monkeypatch, threads, crlf):
    reads = _reads(5000, 1)
    p = str(tmp_path / "r.fq")
    open(p, "wb").write(_fastq_bytes(reads, crlf=crlf))
    monkeypatch.setenv("KAMD_FASTQ_CHUNK", "20000")     # many slices even for this small file
    got, n = _mapped(p, threads)
    assert n == len(reads) and got == reads


def test_plain_reader_rejects_what_is_not_4_line_fastq(tmp_path):
    p = str(tmp_path / "r.fa")
    open(p, "wb").write(b">a\nACGT\nACGT\n>b\nGGGG\n")
    got, _ = _mapped(p, 4)
    assert got is None                                    # the front-end then takes the serial reader:
    got, n = _chunks(p, 1)
    assert got == [b"ACGTACGT", b"GGGG"] and n == 2


def test_multi_line_fastq_and_missing_final_newline(tmp_path):
    p = str(tmp_path / "m.fq")
    open(p, "wb").write(b"@a\nACGT\nAC\n+\nIIII\nII\n@b\nGGGG\n+\nIIII")
    got, n = _chunks(p, 1)
    assert got == [b"ACGTAC", b"GGGG"] and n == 2


@pytest.mark.parametrize("chunk", [1, 777, 100000])
def test_gzip_serial_reader(tmp_path, chunk):
    reads = _reads(3000, 2)
    p = str(tmp_path / "r.fq.gz")
    with gzip.open(p, "wb") as f:
        f.write(_fastq_bytes(reads))
    assert E.lib().io_is_bgzf(p.encode()) == 0
    got, n = _chunks(p, 4, chunk=chunk)                   # not BGZF: the thread count is irrelevant
    assert n == len(reads) and got == reads


@pytest.mark.parametrize("threads", [1, 2, 5, 16])
def test_bgzf_block_parallel_inflate(tmp_path, threads):
    reads = _reads(40000, 3)                              # ~5 MB of FASTQ -> ~80 BGZF blocks, more than the slot ring
    data = _fastq_bytes(reads)
    p = str(tmp_path / "r.fq.gz")
    open(p, "wb").write(_bgzf(data))
    assert gzip.open(p, "rb").read() == data              # the fixture is a valid multi-member gzip file
    assert E.lib().io_is_bgzf(p.encode()) == 1
    got, n = _chunks(p, threads, chunk=5000)
    assert n == len(reads) and got == reads


def test_bgzf_tiny_blocks_and_empty_members(tmp_path):
    reads = _reads(2000, 4, lo=1, hi=40)
    data = _fastq_bytes(reads, tricky_quals=False)
    p = str(tmp_path / "t.fq.gz")
    blob = _bgzf(data, block=97)                          # thousands of members; records straddle member boundaries
    empty = _bgzf(b"")                                    # an EOF member in the middle is legal (concatenated files)
    open(p, "wb").write(blob[:len(blob) - len(empty)] + empty + empty)
    got, n = _chunks(p, 7, chunk=300)
    assert n == len(reads) and got == reads
