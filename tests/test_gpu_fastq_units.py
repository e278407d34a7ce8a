"""kamd_fastq_unit_pack on the GPU (k_fq_count / k_fq_scan / k_fq_fill / k_fq_records / k_fq_pack): a unit of strict 4-line FASTQ text
in HBM must pack to exactly what the host packer makes of the sequences a plain Python parse yields; anything that is not strict is
declined with the index of the first offending record; and the whole front-end on such input (plain / gzip / BGZF, with a late
violation that forces the restart through the general reader) gives the files the host-parsed path gives."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from tests import common
from tests.test_fastq_input import _bgzf, _fastq_bytes, _reads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import kallisto_amd as ka
    c = ka.Context(0)
    yield c
    c.close()


def _expect(ctx, reads):
    w, l, mx = ctx.pack_reads_host(reads)
    return w.cpu().numpy(), l.cpu().numpy(), mx


@pytest.mark.parametrize("crlf", [False, True])
def test_single_unit_equals_host_packer(ctx, crlf):
    reads = _reads(30000, 11, lo=1, hi=260, n_frac=0.05)
    text = _fastq_bytes(reads, crlf=crlf)
    w, l, n, mx, st, bad = ctx.fastq_unit_pack([text], len(reads))
    assert st == 0 and n == len(reads)
    ew, el, emx = _expect(ctx, reads)
    assert mx == emx
    assert np.array_equal(l.cpu().numpy(), el) and np.array_equal(w.cpu().numpy(), ew)


def test_unit_as_a_view_of_a_device_text_buffer(ctx):
    """A unit handed over as a uint8 device tensor: an exactly-sized tensor is copied into a padded one (the packer reads up to 32 bytes behind the
    text), a VIEW of a larger buffer -- what a front-end cuts out of its device text ring, here with junk behind the unit -- is passed as it is
    (no device-to-device copy of the text).  Both give what the bytes give."""
    import torch
    reads = _reads(5000, 17, lo=1, hi=150, n_frac=0.05)
    text = _fastq_bytes(reads)
    ew, el, emx = _expect(ctx, reads)
    exact = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    ring = torch.full((len(text) + 4096,), ord("A"), dtype=torch.uint8, device="cuda")
    ring[:len(text)] = exact
    for t in (exact, ring[:len(text)]):
        w, l, n, mx, st, bad = ctx.fastq_unit_pack([t], len(reads))
        assert st == 0 and n == len(reads) and mx == emx
        assert np.array_equal(l.cpu().numpy(), el) and np.array_equal(w.cpu().numpy(), ew)
    keep, ptrs, nb = ctx._fastq_texts([ring[:len(text)]])
    assert keep[0][0].data_ptr() == ring.data_ptr()          # the view itself went down, not a copy
    keep, ptrs, nb = ctx._fastq_texts([exact])
    assert keep[0][0].data_ptr() != exact.data_ptr() or exact.untyped_storage().nbytes() >= len(text) + 32


def test_paired_unit_interleaves_the_mates(ctx):
    r1, r2 = _reads(20000, 12), _reads(20000, 13, lo=30, hi=101)
    w, l, n, mx, st, _ = ctx.fastq_unit_pack([_fastq_bytes(r1), _fastq_bytes(r2, tricky_quals=False)], len(r1))
    assert st == 0 and n == len(r1)
    inter = [x for pair in zip(r1, r2) for x in pair]
    ew, el, emx = _expect(ctx, inter)
    assert mx == emx and np.array_equal(l.cpu().numpy(), el) and np.array_equal(w.cpu().numpy(), ew)


def test_batch_of_several_units(ctx):
    """kamd_fastq_unit_parse x 3 + kamd_fastq_batch_pack: the batch is the units' reads in order, at the longest read's stride; a unit
    that is declined adds nothing."""
    r1, r2 = _reads(9000, 21, lo=20, hi=90), _reads(9000, 22, lo=40, hi=151)
    cuts = [0, 2500, 2501, 9000]
    for a, b in zip(cuts[:-1], cuts[1:]):
        if a == 2500:   # a bad unit in between: declined, the batch is unchanged
            st, bad, _ = ctx.fastq_unit_parse([b"@x\nAC\n+\nI\n", b"@x\nAC\n+\nII\n"], 1)
            assert st == 1 and bad == 0
        st, _, _ = ctx.fastq_unit_parse([_fastq_bytes(r1[a:b]), _fastq_bytes(r2[a:b], tricky_quals=False)], b - a)
        assert st == 0
    w, l, n, mx, st, _ = ctx.fastq_batch_pack()
    assert st == 0 and n == 9000
    inter = [x for pair in zip(r1, r2) for x in pair]
    ew, el, emx = _expect(ctx, inter)
    assert mx == emx and np.array_equal(l.cpu().numpy(), el) and np.array_equal(w.cpu().numpy(), ew)
    assert ctx.fastq_batch_pack()[2] == 0                 # nothing left


def test_every_byte_value_as_a_base(ctx):
    """lower case is read like upper case, everything that is not ACGT is a masked base with code 0 -- for every byte value a FASTQ
    sequence line can hold, at every position of a 32-base group"""
    alphabet = bytes(b for b in range(1, 256) if b not in (10, 13))        # no line ends
    reads = []
    for shift in range(33):
        body = (b"ACGT" * 9)[:shift] + alphabet
        if body[0:1] in (b"@", b"+", b">"):                                 # (not as the first character: kseq would take it for a header)
            body = b"A" + body
        reads.append(body)
    reads += [b"acgtnACGTN" * 7, b"a", b"n", b"tT" * 16, b"gG" * 16 + b"c"]
    text = _fastq_bytes(reads, tricky_quals=False)
    w, l, n, mx, st, _ = ctx.fastq_unit_pack([text], len(reads))
    assert st == 0 and n == len(reads)
    ew, el, emx = _expect(ctx, reads)
    assert mx == emx and np.array_equal(l.cpu().numpy(), el) and np.array_equal(w.cpu().numpy(), ew)


def test_tiny_and_unaligned_sizes(ctx):
    for n in [1, 2, 3, 17, 255, 256, 257]:
        reads = _reads(n, 100 + n, lo=1, hi=40)
        text = _fastq_bytes(reads)
        w, l, k, mx, st, _ = ctx.fastq_unit_pack([text], n)
        assert st == 0 and k == n
        ew, el, emx = _expect(ctx, reads)
        assert mx == emx and np.array_equal(w.cpu().numpy(), ew) and np.array_equal(l.cpu().numpy(), el)


def test_declines_records_that_are_not_strict(ctx):
    reads = _reads(5000, 14)
    base = _fastq_bytes(reads, tricky_quals=False)
    recs = base.split(b"@r")[1:]                          # "N some comment\nSEQ\n+\nQUAL\n"
    def with_record(i, rec):
        return b"".join(b"@r" + (rec if j == i else r) for j, r in enumerate(recs))
    name, seq, plus, qual = recs[4321].split(b"\n")[:4]
    cases = [
        name + b"\n" + seq + b"\nAC\n" + plus + b"\n" + qual + b"II\n",       # 6 lines: shifts everything behind it
        name + b"\n" + seq + b"\n" + plus + b"\n" + qual[:-1] + b"\n",        # quality too short
        name + b"\n" + seq + b"\n-\n" + qual + b"\n",                          # third line does not start with '+'
        name + b"\n\n" + plus + b"\n\n",                                       # empty sequence
        name + b"\n>" + seq[1:] + b"\n" + plus + b"\n" + qual + b"\n",        # sequence line starts with '>'
    ]
    for k, rec in enumerate(cases):
        text = with_record(4321, rec)
        _, _, _, _, st, bad = ctx.fastq_unit_pack([text], len(reads))
        assert st == 1 and bad == 4321, (k, st, bad)
    # fewer lines than the caller promised
    _, _, _, _, st, _ = ctx.fastq_unit_pack([base], len(reads) + 1)
    assert st == 2
    # junk in front of the first header
    _, _, _, _, st, bad = ctx.fastq_unit_pack([b"x" + base], len(reads))
    assert st == 1 and bad == 0


def test_read_longer_than_the_packed_layout(ctx):
    reads = [b"ACGT" * 5, b"A" * 70000]
    _, _, _, mx, st, _ = ctx.fastq_unit_pack([_fastq_bytes(reads, tricky_quals=False)], 2)
    assert st == 3 and mx == 70000


EXE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kallisto_amd", "kallisto_amd_quant")


def _run_cli(idx, files, out, extra=(), env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([EXE, "quant", "-i", idx, "-o", out, "--plaintext", "-t", "4", "--verbose", *extra, *files], env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return open(os.path.join(out, "abundance.tsv"), "rb").read(), p.stderr.decode()


@pytest.mark.parametrize("kind", ["plain", "gzip", "bgzf"])
def test_front_end_device_parser_equals_general_reader(tmp_path, kind):
    meta, idx_path, r1, r2 = common.load_case("human_pe")
    files = []
    for i, reads in enumerate((r1, r2)):
        data = _fastq_bytes(reads, tricky_quals=True)
        p = str(tmp_path / f"r_{i}.fq") + ("" if kind == "plain" else ".gz")
        if kind == "plain":
            open(p, "wb").write(data)
        elif kind == "gzip":
            with gzip.open(p, "wb", compresslevel=3) as f:
                f.write(data)
        else:
            open(p, "wb").write(_bgzf(data, block=20000))
        files.append(p)
    small = {"KAMD_FQ_UNIT_BYTES": "300000", "KAMD_FQ_BATCH_ITEMS": "3000"}                       # several units even for these small files
    dev_tsv, dev_err = _run_cli(idx_path, files, str(tmp_path / "dev"), env=small)
    host_tsv, host_err = _run_cli(idx_path, files, str(tmp_path / "host"), env={"KAMD_HOST_PARSE": "1"})
    assert "device parser: 0 units" not in dev_err and "device parser: 0 units" in host_err
    assert dev_tsv == host_tsv


def test_front_end_restarts_with_the_general_reader_on_a_late_violation(tmp_path):
    meta, idx_path, r1, r2 = common.load_case("human_pe")
    d1, d2 = _fastq_bytes(r1, tricky_quals=False), _fastq_bytes(r2, tricky_quals=False)
    # the last record of both files in FASTA-like multi-line form: kseq reads it, the strict parser must decline the unit
    def tail_multiline(data):
        recs = data.split(b"\n")
        name, seq, plus, qual = recs[-5], recs[-4], recs[-3], recs[-2]
        h = len(seq) // 2
        return b"\n".join(recs[:-5]) + b"\n" + name + b"\n" + seq[:h] + b"\n" + seq[h:] + b"\n" + plus + b"\n" + qual[:h] + b"\n" + qual[h:] + b"\n"
    a, b = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
    open(a, "wb").write(tail_multiline(d1)); open(b, "wb").write(tail_multiline(d2))
    ref_a, ref_b = str(tmp_path / "ra.fq"), str(tmp_path / "rb.fq")
    open(ref_a, "wb").write(d1); open(ref_b, "wb").write(d2)
    got, err = _run_cli(idx_path, [a, b], str(tmp_path / "o1"), env={"KAMD_FQ_UNIT_BYTES": "300000", "KAMD_FQ_BATCH_ITEMS": "3000"})
    want, _ = _run_cli(idx_path, [ref_a, ref_b], str(tmp_path / "o2"))
    assert "general FASTA/FASTQ reader" in err
    assert got == want
