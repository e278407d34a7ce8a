"""Golden fixtures for the bulk `bus` / `quant-tcc` consumers, produced by the UNMODIFIED reference
(oracle/_ref/kallisto, built by `make -C oracle ref`).  Run in the build container:  python tests/golden/make_bus_tcc.py

For every case below the reads of an existing fixture are split into samples and run through

    kallisto bus -x bulk -t 1 [--paired] [variant flags] -i index.idx  sample files...
    (bustools is not available offline: `bustools sort` + `bustools count` are restated here -- collapse identical records,
     one matrix row per barcode, one column per class of matrix.ec)
    kallisto quant-tcc -t 1 -i index.idx -e matrix.ec [-f flens.txt | -l L -s S] [--matrix-to-files] [-b B --plaintext] tcc.mtx

and tests/golden/bus_tcc/<case>/ keeps
    case.json                what was run
    bus_expected.txt         the BUS file as sorted lines "barcode<TAB>count<TAB>t1,t2,..." (ids resolved through matrix.ec)
    flens.txt, run_info.json the reference's files (paired cases; run_info without start_time / call)
    matrix.ec, tcc.mtx       the reference's classes and the count matrix made from its BUS file: the input of quant-tcc
    tcc_out/                 the reference's quant-tcc output (matrix.*.mtx, matrix.fld.tsv, abundance_N.tsv, bs_abundance_N_B.tsv)
"""
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import common  # noqa: E402

KALLISTO = os.path.join(ROOT, "oracle", "_ref", "kallisto")
OUT = os.path.join(common.GOLDEN, "bus_tcc")
BUS_DTYPE = np.dtype([("bc", "<u8"), ("umi", "<u8"), ("ec", "<i4"), ("count", "<u4"), ("flags", "<u4"), ("pad", "<u4")])

# name: (fixture, sample boundaries as fractions, bus flags, quant-tcc flags)
CASES = {
    "human_pe_2": ("human_pe", [0.0, 0.6, 1.0], ["--paired"], ["--matrix-to-files"]),
    "human_pe_fr": ("human_pe", [0.0, 0.5, 1.0], ["--paired", "--fr-stranded"], []),
    "ref_test_pe_boot": ("ref_test_pe", [0.0, 0.3, 0.7, 1.0], ["--paired"], ["--matrix-to-files", "-b", "2", "--seed", "11", "--plaintext"]),
    "yeast_se_2": ("yeast_se", [0.0, 0.4, 1.0], [], ["-l", "200", "-s", "20"]),
    "yeast_se_noeff": ("yeast_se", [0.0, 1.0], ["--rf-stranded"], []),
    "mosaic_pe_union": ("mosaic_pe", [0.0, 0.5, 1.0], ["--paired", "--union"], []),
    "dlist_pe_nojump": ("dlist_pe", [0.0, 0.5, 1.0], ["--paired", "--no-jump"], []),
    # -g: gene-level sums ("@file" = a file of the case's directory: genemap.txt, written below)
    "human_pe_genes": ("human_pe", [0.0, 0.6, 1.0], ["--paired"], ["--matrix-to-files", "-g", "@genemap.txt"]),
    # a directory per sample, with gene-level files of the bootstrap replicates in it
    "ref_test_pe_dirs_boot": ("ref_test_pe", [0.0, 0.45, 1.0], ["--paired"], ["--matrix-to-directories", "-g", "@genemap.txt", "-b", "2", "--seed", "7", "--plaintext"]),
}


def write_genemap(path, names):
    """transcript -> gene: three transcripts per gene, every other gene with a common name, every tenth transcript in no gene, the lines
    not in transcript order (genes are numbered in order of first appearance in the file)."""
    lines = []
    for i, n in enumerate(names):
        if i % 10 == 9:
            continue
        g = i // 3
        lines.append("%s\tG%04d%s" % (n, g, "\tgene%d" % g if g % 2 == 0 else ""))
    lines = lines[len(lines) // 2:] + lines[:len(lines) // 2]
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n\n")


def write_fastq(path, reads):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))


def read_bus(path):
    b = open(path, "rb").read()
    assert b[:4] == b"BUS\0"
    ver, bclen, umilen, tlen = struct.unpack("<IIII", b[4:20])
    return (ver, bclen, umilen), np.frombuffer(b[20 + tlen:], dtype=BUS_DTYPE)


def read_ec(path):
    ecs = []
    for i, line in enumerate(open(path)):
        e, trs = line.split()
        assert int(e) == i
        ecs.append(tuple(sorted(int(x) for x in trs.split(","))))
    return ecs


def collapse(rec):
    """`bustools sort`: identical (barcode, umi, ec, flags) records are merged, counts summed."""
    d = {}
    for r in rec:
        k = (int(r["bc"]), int(r["umi"]), int(r["ec"]), int(r["flags"]))
        d[k] = d.get(k, 0) + int(r["count"])
    return d


def bus_lines(rec, ecs):
    out = {}
    for (bc, umi, ec, flags), n in collapse(rec).items():
        assert umi == 2 ** 64 - 1 and flags == 0
        k = (bc, ecs[ec])
        out[k] = out.get(k, 0) + n
    return ["%d\t%d\t%s" % (bc, n, ",".join(map(str, s))) for (bc, s), n in sorted(out.items())]


def write_tcc(path, rec, n_ecs, n_samples):
    """`bustools count` on a bulk file: rows = barcodes (samples), columns = classes, values = reads."""
    d = {}
    for (bc, umi, ec, flags), n in collapse(rec).items():
        d[(bc, ec)] = d.get((bc, ec), 0) + n
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate integer general\n")
        f.write("%d\t%d\t%d\n" % (n_samples, n_ecs, len(d)))
        for (bc, ec), n in sorted(d.items()):
            f.write("%d\t%d\t%d\n" % (bc + 1, ec + 1, n))


def main():
    if not os.path.exists(KALLISTO):
        sys.exit("oracle/_ref is not built: run `make -C oracle ref` in the build container")
    only = sys.argv[1:]
    for name, (fixture, cuts, bus_flags, tcc_flags) in CASES.items():
        if only and name not in only:
            continue
        meta, idx, r1, r2 = common.load_case(fixture)
        paired = "--paired" in bus_flags
        n = len(r1)
        dst = os.path.join(OUT, name)
        shutil.rmtree(dst, ignore_errors=True)
        os.makedirs(dst)
        with tempfile.TemporaryDirectory() as tmp:
            files = []
            for s in range(len(cuts) - 1):
                a, b = int(round(cuts[s] * n)), int(round(cuts[s + 1] * n))
                f1 = os.path.join(tmp, "s%d_1.fq" % s)
                write_fastq(f1, r1[a:b])
                files.append(f1)
                if paired:
                    f2 = os.path.join(tmp, "s%d_2.fq" % s)
                    write_fastq(f2, r2[a:b])
                    files.append(f2)
            bus_out = os.path.join(tmp, "bus")
            p = subprocess.run([KALLISTO, "bus", "-x", "bulk", "-t", "1", "-i", idx, "-o", bus_out, *bus_flags, *files],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert p.returncode == 0, p.stderr.decode()
            hdr, rec = read_bus(os.path.join(bus_out, "output.bus"))
            ecs = read_ec(os.path.join(bus_out, "matrix.ec"))
            with open(os.path.join(dst, "bus_expected.txt"), "w") as f:
                f.write("\n".join(bus_lines(rec, ecs)) + "\n")
            info = json.load(open(os.path.join(bus_out, "run_info.json")))
            for k in ("start_time", "call"):
                info.pop(k)
            json.dump(info, open(os.path.join(dst, "run_info.json"), "w"), indent=1)
            for fn in ("matrix.cells", "matrix.sample.barcodes", "transcripts.txt"):
                shutil.copy(os.path.join(bus_out, fn), dst)
            if paired:
                shutil.copy(os.path.join(bus_out, "flens.txt"), dst)
            # only the classes that occur go into the quant-tcc input (renumbered): the fixture stays small and does not depend on
            # the reference's numbering of the index's own classes
            used = sorted({int(e) for e in rec["ec"]})
            renum = {e: i for i, e in enumerate(used)}
            with open(os.path.join(dst, "matrix.ec"), "w") as f:
                for e in used:
                    f.write("%d\t%s\n" % (renum[e], ",".join(map(str, ecs[e]))))
            rec2 = rec.copy()
            rec2["ec"] = [renum[int(e)] for e in rec["ec"]]
            n_samples = len(cuts) - 1
            write_tcc(os.path.join(dst, "tcc.mtx"), rec2, len(used), n_samples)
            tcc_out = os.path.join(dst, "tcc_out")
            if "@genemap.txt" in tcc_flags:
                write_genemap(os.path.join(dst, "genemap.txt"), [l.strip() for l in open(os.path.join(bus_out, "transcripts.txt"))])
            tcc_run_flags = [os.path.join(dst, a[1:]) if a.startswith("@") else a for a in tcc_flags]
            fld_args = []
            if "-l" not in tcc_flags and paired:
                fld_args = ["-f", os.path.join(dst, "flens.txt")]
            p = subprocess.run([KALLISTO, "quant-tcc", "-t", "1", "-i", idx, "-e", os.path.join(dst, "matrix.ec"), "-o", tcc_out, *fld_args,
                                *tcc_run_flags, os.path.join(dst, "tcc.mtx")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert p.returncode == 0, p.stderr.decode()
        json.dump({"fixture": fixture, "cuts": cuts, "bus_flags": bus_flags, "tcc_flags": tcc_flags, "bus_header": list(hdr),
                   "n_records_reference": int(len(rec)), "fld_file": bool(fld_args),
                   "reference": "pachterlab/kallisto v0.51.1, oracle/_ref/kallisto (unmodified sources), -t 1"},
                  open(os.path.join(dst, "case.json"), "w"), indent=1)
        print(name, "samples", len(cuts) - 1, "records", len(rec), "classes used", len(used),
              sorted(os.path.relpath(os.path.join(d, f), tcc_out) for d, _, fs in os.walk(tcc_out) for f in fs))


if __name__ == "__main__":
    main()
