"""Input side of the device FASTQ parser on the CPU (kallisto_amd/csrc/kamd_textsource.h + kamd_fq_core.h through tests/emu/fq_emu.cpp):
host threads move the text of a file into a ring and count its newlines (TextSource: plain / gzip / BGZF), UnitCutter cuts the
stream into units of whole 4-line records, and the per-unit parse (restated serially over the device's own host/device functions)
must return exactly the sequences kseq_read returns (src/kseq.h:174-215) -- or decline, so that the front-end takes the general
reader.  The kernels proper are checked on the GPU (tests/test_gpu_fastq_units.py)."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

from tests import emu_binding as E
from tests.test_fastq_input import _bgzf, _fastq_bytes, _reads


def _units(p0, p1=None, ring=1 << 20, target=1 << 16, mx=None, threads=4, blk=1 << 14, cap=1 << 27):
    L = E.lib()
    L.io_units.restype = C.c_int64
    buf = C.create_string_buffer(cap)
    n_rec, n_units, max_len = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
    r = L.io_units(p0.encode(), (p1 or "").encode(), C.c_uint64(ring), C.c_uint64(target), C.c_uint64(mx or ring // 2), int(threads),
                   C.c_uint64(blk), buf, C.c_uint64(cap), C.byref(n_rec), C.byref(n_units), C.byref(max_len))
    if r < 0:
        return r, None, 0, 0
    return 0, buf.raw[:r].split(b"\n")[:-1], n_rec.value, n_units.value


def test_newline_counting_primitives():
    L = E.lib()
    L.io_count_newlines.restype = C.c_uint64
    L.io_after_kth_newline.restype = C.c_uint64
    rng = np.random.default_rng(0)
    for n in [0, 1, 15, 16, 17, 63, 64, 65, 1000, 4099]:
        a = rng.integers(0, 256, n, dtype=np.uint8)
        a[rng.random(n) < 0.1] = 10
        b = a.tobytes()
        pos = [i for i, c in enumerate(b) if c == 10]
        assert L.io_count_newlines(b, C.c_uint64(n)) == len(pos)
        for k in [1, 2, len(pos) // 2, len(pos)]:
            if 1 <= k <= len(pos):
                assert L.io_after_kth_newline(b, C.c_uint64(n), C.c_uint64(k)) == pos[k - 1] + 1
        assert L.io_after_kth_newline(b, C.c_uint64(n), C.c_uint64(len(pos) + 1)) == n
    # the wide (AVX2) count, the SSE2 count and the fused copy + count agree, at every alignment of source and destination
    L.io_copy_count_newlines.restype = C.c_uint64
    L.io_count_newlines_sse2.restype = C.c_uint64
    a = rng.integers(0, 256, 70000, dtype=np.uint8)
    a[rng.random(a.size) < 0.05] = 10
    src = C.create_string_buffer(a.tobytes(), a.size)
    dst = C.create_string_buffer(a.size + 64)
    for so in (0, 1, 7, 31, 33):
        for do in (0, 3, 32, 45):
            for n in (0, 1, 31, 127, 128, 129, 4097, 60000):
                want = int((a[so:so + n] == 10).sum())
                sp = C.cast(C.addressof(src) + so, C.c_char_p); dp = C.cast(C.addressof(dst) + do, C.c_void_p)
                assert L.io_count_newlines(sp, C.c_uint64(n)) == want and L.io_count_newlines_sse2(sp, C.c_uint64(n)) == want
                C.memset(dst, 0, a.size + 64)
                assert L.io_copy_count_newlines(dp, sp, C.c_uint64(n)) == want
                assert dst.raw[do:do + n] == a[so:so + n].tobytes() and dst.raw[do + n:do + n + 8] == bytes(8)


@pytest.mark.parametrize("threads", [1, 3, 8])
@pytest.mark.parametrize("crlf", [False, True])
def test_plain_units_single(tmp_path, threads, crlf):
    reads = _reads(20000, 1)
    p = str(tmp_path / "r.fq")
    open(p, "wb").write(_fastq_bytes(reads, crlf=crlf))
    # a ring of 256 KiB for ~3.5 MB of text: the readers wait for the consumer, blocks and units wrap around the ring's end
    st, got, n, nu = _units(p, ring=1 << 18, target=1 << 15, threads=threads, blk=5000)
    assert st == 0 and n == len(reads) and got == reads and nu > 20


@pytest.mark.parametrize("kinds", [("plain", "plain"), ("gzip", "bgzf"), ("bgzf", "gzip"), ("bgzf", "bgzf")])
def test_paired_units_all_sources(tmp_path, kinds):
    r1, r2 = _reads(15000, 2), _reads(15000, 3, lo=30, hi=200)      # the mates' files differ in bytes per record
    paths = []
    for i, (reads, kind) in enumerate(zip((r1, r2), kinds)):
        data = _fastq_bytes(reads, tricky_quals=(i == 0))
        p = str(tmp_path / f"r{i}.fq") + ("" if kind == "plain" else ".gz")
        if kind == "plain":
            open(p, "wb").write(data)
        elif kind == "gzip":
            with gzip.open(p, "wb", compresslevel=4) as f:
                f.write(data)
        else:
            open(p, "wb").write(_bgzf(data, block=30000))
        paths.append(p)
    st, got, n, nu = _units(paths[0], paths[1], ring=1 << 19, target=1 << 16, threads=5, blk=1 << 13)
    assert st == 0 and n == len(r1) and nu > 5
    assert got[0::2] == r1 and got[1::2] == r2


def test_missing_final_newline_and_trailing_blank_lines(tmp_path):
    reads = _reads(300, 5)
    data = _fastq_bytes(reads)
    p = str(tmp_path / "a.fq")
    open(p, "wb").write(data[:-1])                       # no newline behind the last quality line
    st, got, n, _ = _units(p, target=4096)
    assert st == 0 and got == reads
    open(p, "wb").write(data + b"\n\r\n  \n")              # blank lines at the end are junk kseq skips
    st, got, n, _ = _units(p, target=4096)
    assert st == 0 and got == reads
    with gzip.open(p + ".gz", "wb") as f:
        f.write(data[:-1])
    st, got, n, _ = _units(p + ".gz", target=4096)
    assert st == 0 and got == reads
    open(p + ".b.gz", "wb").write(_bgzf(data[:-1], block=5000))
    st, got, n, _ = _units(p + ".b.gz", target=4096)
    assert st == 0 and got == reads


def test_multi_member_gzip_is_one_text(tmp_path):
    reads = _reads(900, 6)
    data = _fastq_bytes(reads)
    cut = data.index(b"\n", len(data) // 2) + 1           # anywhere, even inside a record
    p = str(tmp_path / "m.fq.gz")
    with open(p, "wb") as f:
        f.write(gzip.compress(data[:cut]) + gzip.compress(data[cut:]) + b"\0" * 100)
    st, got, n, _ = _units(p, target=8192)
    assert st == 0 and got == reads


@pytest.mark.parametrize("text", [
    b">a\nACGT\n>b\nGG\n",                                  # FASTA
    b"@a\nACGT\nAC\n+\nIIII\nII\n@b\nGGGG\n+\nIIII\n",      # sequence and quality over two lines
    b"@a\nACGT\n+\nIIII\n\n@b\nGGGG\n+\nIIII\n",            # an empty line between two records
    b"@a\nACGT\n+\nIII\n@b\nGGGG\n+\nIIII\n",               # quality shorter than the sequence
    b"@a\nACGT\n+\nIIIII\n@b\nGGGG\n+\nIIII\n",             # ... longer
    b"@a\n\n+\n\n@b\nGGGG\n+\nIIII\n",                      # empty sequence
    b"@a\nACGT\n+\nIIII\n@b\nGGGG\n",                       # the file ends inside a record
    b"@a\nACGT\n+\nIIII\nxyz",                               # junk behind the last record
])
def test_declines_what_is_not_strict_4_line_fastq(tmp_path, text):
    p = str(tmp_path / "x.fq")
    open(p, "wb").write(text)
    st, got, _, _ = _units(p, target=64)
    assert st in (-1, -11), st                            # the cutter (line counts, tail) or the record check says no


def test_declines_late_violation_and_mate_count_mismatch(tmp_path):
    reads = _reads(4000, 7)
    data = bytearray(_fastq_bytes(reads, tricky_quals=False))
    p = str(tmp_path / "late.fq")
    i = data.rindex(b"\n+\n", 0, len(data) - 400)          # the '+' of a record near the end becomes a base: multi-line record
    data[i + 1] = ord("A")
    open(p, "wb").write(bytes(data))
    st, _, _, _ = _units(p, target=8192)
    assert st == -11
    a, b = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
    open(a, "wb").write(_fastq_bytes(reads))
    open(b, "wb").write(_fastq_bytes(reads[:-3]))
    assert _units(a, b, target=8192)[0] == -2
    assert _units(b, a, target=8192)[0] == -2


def test_record_larger_than_a_unit(tmp_path):
    reads = [b"ACGT" * 10, b"A" * 50000, b"GGCC" * 5]
    p = str(tmp_path / "big.fq")
    open(p, "wb").write(_fastq_bytes(reads, tricky_quals=False))
    st, got, _, _ = _units(p, ring=1 << 20, target=1024, mx=1 << 19, blk=4096)
    assert st == 0 and got == reads                       # the cutter widens the unit until it holds a record
    st, _, _, _ = _units(p, ring=1 << 16, target=1024, mx=1 << 15, blk=4096)
    assert st == -4                                       # ... and gives up when the ring cannot hold one


def test_carriage_return_rule_of_kseq(tmp_path):
    # ks_getuntil2 drops a trailing '\r' only from a line of more than one character (src/kseq.h:137)
    p = str(tmp_path / "cr.fq")
    open(p, "wb").write(b"@a\r\nA\r\n+\r\nI\r\n@b\r\n\r\n+\r\n\r\n")
    st, got, _, _ = _units(p, target=16)
    assert st == 0 and got == [b"A", b"\r"]


# ---- block-parallel inflate of ordinary gzip files (kallisto_amd/csrc/kamd_pargzip.h) ----------------------------------------
def _pargzip_env(monkeypatch, chunk_kb=16):
    monkeypatch.setenv("KAMD_PARGZIP_MIN_KB", "0")
    monkeypatch.setenv("KAMD_PARGZIP_CHUNK_KB", str(chunk_kb))


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("chunk_kb", [4, 64])
def test_parallel_gzip_equals_the_text(tmp_path, monkeypatch, level, chunk_kb):
    """A gzip file cut into chunks of 4 / 64 KiB of compressed bytes (dozens to hundreds of chunks: block starts found inside the
    stream, symbols with markers, window resolution, in-order delivery, CRC-32 and length of the member checked) gives exactly the
    reads of the text; the ring is smaller than the text, so the decoder also waits for the consumer."""
    _pargzip_env(monkeypatch, chunk_kb)
    reads = _reads(30000, 11 + level)
    data = _fastq_bytes(reads, tricky_quals=True)
    p = str(tmp_path / "r.fq.gz")
    open(p, "wb").write(gzip.compress(data, level))
    st, got, n, nu = _units(p, ring=1 << 19, target=1 << 16, threads=6, blk=1 << 13)
    assert st == 0 and n == len(reads) and got == reads and nu > 10


def test_parallel_gzip_stored_fixed_blocks_members_and_junk(tmp_path, monkeypatch):
    """Blocks the block finder cannot recognise (stored, fixed Huffman: level 0 and Z_FIXED streams), several members, zero padding
    and junk behind the last member: everything the parallel path does not cover goes through its serial decoder -- same text."""
    import zlib
    _pargzip_env(monkeypatch, 8)
    reads = _reads(12000, 21)
    data = _fastq_bytes(reads)
    a, b, c = data[:len(data) // 3], data[len(data) // 3:2 * len(data) // 3], data[2 * len(data) // 3:]
    def member(raw, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
        co = zlib.compressobj(level, zlib.DEFLATED, 31, 9, strategy)
        return co.compress(raw) + co.flush()
    blob = member(a, 0) + member(b, 6, zlib.Z_FIXED) + b"\0" * 37 + member(c, 6) + b"not a gzip member"
    p = str(tmp_path / "mix.fq.gz")
    open(p, "wb").write(blob)
    st, got, n, _ = _units(p, ring=1 << 19, target=1 << 15, threads=4)
    assert st == 0 and got == reads
    # a text without the final newline, and the same through the parallel path as a paired input against a plain mate
    open(p, "wb").write(gzip.compress(data[:-1], 6))
    st, got, n, _ = _units(p, target=1 << 15, threads=4)
    assert st == 0 and got == reads


def test_parallel_gzip_reports_corruption(tmp_path, monkeypatch):
    """A flipped byte in the middle of the deflate data, a wrong CRC in the trailer, a truncated file: an error, never a clean run."""
    _pargzip_env(monkeypatch, 8)
    reads = _reads(12000, 31)
    blob = bytearray(gzip.compress(_fastq_bytes(reads), 6))
    p = str(tmp_path / "bad.fq.gz")
    for what in ("flip", "crc", "truncate"):
        bad = bytearray(blob)
        if what == "flip":
            bad[len(bad) // 2] ^= 0x5A
        elif what == "crc":
            bad[-6] ^= 0xFF
        else:
            bad = bad[:len(bad) * 2 // 3]
        open(p, "wb").write(bytes(bad))
        st, got, _, _ = _units(p, target=1 << 15, threads=4)
        # UnitCutter::IO_ERROR; a flipped byte may garble the text before the decoder meets an invalid code or the CRC: then the record
        # check declines first (-11) and the general reader, which the front-end turns to, fails on the same file
        assert st == -3 or (what == "flip" and st == -11), (what, st)


def test_parallel_gzip_flush_points_and_many_members(tmp_path, monkeypatch):
    """What parallel compressors write: empty stored blocks at sync / full flush points in the middle of the stream (pigz), and files made
    of hundreds of small members (concatenated pieces): the parallel reader walks through all of it."""
    import zlib
    _pargzip_env(monkeypatch, 8)
    reads = _reads(15000, 41)
    data = _fastq_bytes(reads)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    blob = b""
    step = 70000
    for i, a in enumerate(range(0, len(data), step)):
        blob += co.compress(data[a:a + step]) + co.flush(zlib.Z_FULL_FLUSH if i % 3 == 0 else zlib.Z_SYNC_FLUSH)
    blob += co.flush()
    p = str(tmp_path / "pigz.fq.gz")
    open(p, "wb").write(blob)
    st, got, n, _ = _units(p, ring=1 << 19, target=1 << 15, threads=5)
    assert st == 0 and got == reads
    # 300 members, cut anywhere (inside records too)
    cuts = sorted(set(np.random.default_rng(3).integers(1, len(data) - 1, 299).tolist()))
    blob = b"".join(gzip.compress(data[a:b], 4) for a, b in zip([0] + cuts, cuts + [len(data)]))
    open(p, "wb").write(blob)
    st, got, n, _ = _units(p, ring=1 << 19, target=1 << 15, threads=5)
    assert st == 0 and got == reads


def _random_gzip_case(seed):
    """A random FASTQ text (constant / random / narrow qualities, short or long names, read lengths from a random range) compressed in a random
    way: level, strategy, flush points, several members, a small window."""
    import zlib
    rng = np.random.default_rng(seed)
    n = int(rng.integers(200, 6000))
    lo = int(rng.integers(1, 60)); hi = lo + int(rng.integers(1, 250))
    acgt = np.frombuffer(b"ACGTN", np.uint8)
    qmode = int(rng.integers(0, 3))
    reads, parts = [], []
    for i in range(n):
        L = int(rng.integers(lo, hi))
        s = bytes(acgt[rng.integers(0, 5 if rng.random() < 0.02 else 4, L)])
        q = b"I" * L if qmode == 0 else bytes(rng.integers(33, 74, L, dtype=np.uint8)) if qmode == 1 else bytes(rng.integers(35, 40, L, dtype=np.uint8))
        name = b"@r%d" % i if rng.random() < 0.5 else b"@inst:%d:%d:%d %d:N:0" % (rng.integers(1, 99), rng.integers(1, 9999), rng.integers(1, 99999), rng.integers(1, 3))
        reads.append(s); parts.append(name + b"\n" + s + b"\n+\n" + q + b"\n")
    data = b"".join(parts)
    level = int(rng.integers(1, 10))
    strat = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE][int(rng.integers(0, 4))] if rng.random() < 0.3 else zlib.Z_DEFAULT_STRATEGY
    mode = int(rng.integers(0, 4))
    if mode == 0:
        co = zlib.compressobj(level, zlib.DEFLATED, 31, int(rng.integers(1, 10)), strat); blob = co.compress(data) + co.flush()
    elif mode == 1:
        co = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strat); blob = b""; step = int(rng.integers(1000, 200000))
        for a in range(0, len(data), step):
            blob += co.compress(data[a:a + step]) + co.flush([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_PARTIAL_FLUSH][int(rng.integers(0, 3))])
        blob += co.flush()
    elif mode == 2:
        k = int(rng.integers(2, 40)); cuts = sorted(set(rng.integers(1, len(data) - 1, k).tolist()))
        blob = b"".join(gzip.compress(data[a:b], int(rng.integers(0, 10))) for a, b in zip([0] + cuts, cuts + [len(data)]))
    else:
        co = zlib.compressobj(level, zlib.DEFLATED, 16 + int(rng.integers(9, 16)), 8, strat); blob = co.compress(data) + co.flush()
    knobs = dict(chunk_kb=int(rng.choice([1, 2, 3, 4, 8, 16, 64, 256])), threads=int(rng.integers(1, 9)), ring=1 << int(rng.integers(18, 23)),
                 target=1 << int(rng.integers(13, 17)), blk=1 << int(rng.integers(12, 15)))
    return reads, data, blob, knobs


@pytest.mark.parametrize("seed", range(7000, 7024))
def test_parallel_gzip_random_streams(tmp_path, monkeypatch, seed):
    """Two dozen random streams and reader shapes: exactly the text's reads.  (Offline the same generator ran 3 000+ larger cases without a
    difference, 580 of them and 2 257 corrupted files -- flipped / overwritten / deleted bytes, truncation: an error every time, never a wrong
    text -- with the library built under AddressSanitizer + UBSan: no report.)"""
    reads, data, blob, k = _random_gzip_case(seed)
    _pargzip_env(monkeypatch, k["chunk_kb"])
    p = str(tmp_path / "f.fq.gz")
    open(p, "wb").write(blob)
    st, got, n, _ = _units(p, ring=k["ring"], target=k["target"], threads=k["threads"], blk=k["blk"], cap=max(1 << 24, 2 * len(data)))
    assert st == 0 and got == reads, (seed, st, k)
