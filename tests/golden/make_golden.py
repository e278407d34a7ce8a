"""Generate the golden fixtures under tests/golden/ with the UNMODIFIED reference (oracle/_ref, built from
/root/reference by oracle/Makefile).  Run in the build container only:

    python tests/golden/make_golden.py

Each case directory holds
    index.idx            built by `kallisto index` (reference binary)
    reads_1.txt.gz [reads_2.txt.gz]   one read per line
    expected_<variant>.txt            output of oracle/_ref/dump_ec quant ... (NPROC / EC / FLEN / TR [/ BS] lines)
    case.json                         how the case was made (seeds, options per variant)
The fixtures are small on purpose (a few hundred kB each); they pin the oracle (tests/test_oracle_golden.py) and the
HIP path (tests/test_gpu_parity.py) without needing /root/reference at test time.
"""
from __future__ import annotations

import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from kallisto_amd import synth  # noqa: E402
sys.path.insert(0, HERE)
import h5dump_tools  # noqa: E402

ONLY = set(sys.argv[1:])   # optional: only (re)generate the named cases
REF = os.path.join(ROOT, "oracle", "_ref")
KALLISTO = os.path.join(REF, "kallisto")
DUMP = os.path.join(REF, "dump_ec")
KALLISTO_H5 = os.path.join(REF, "kallisto_h5")
H5DUMP = "/opt/conda/bin/h5dump"
H5_VARIANTS = {("ref_test_pe", "pe_boot"), ("yeast_se", "se"), ("dlist_pe", "pe")}


def write_lines(path, reads):
    with gzip.open(path, "wb", compresslevel=9) as f:
        for r in reads:
            f.write(bytes(r) + b"\n")


def write_fastq(path, reads):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            s = bytes(r)
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)))


def make_case(name, fasta_path, reads1, reads2, variants, k=31, note="", index_args=()):
    if ONLY and name not in ONLY:
        return
    d = os.path.join(HERE, name)
    os.makedirs(d, exist_ok=True)
    idx = os.path.join(d, "index.idx")
    subprocess.check_call([KALLISTO, "index", "-k", str(k), "-i", idx, *index_args, fasta_path], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    write_lines(os.path.join(d, "reads_1.txt.gz"), reads1)
    if reads2 is not None:
        write_lines(os.path.join(d, "reads_2.txt.gz"), reads2)
    with tempfile.TemporaryDirectory() as tmp:
        f1 = os.path.join(tmp, "r1.fq")
        write_fastq(f1, reads1)
        files = [f1]
        if reads2 is not None:
            f2 = os.path.join(tmp, "r2.fq")
            write_fastq(f2, reads2)
            files.append(f2)
        for vname, extra in variants.items():
            vfiles = files[:1] if "--single" in extra else files   # single-end variants of a paired case use mate 1 only
            out = subprocess.run([DUMP, "quant", idx, "1", *extra, *vfiles], check=True, stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL).stdout
            with open(os.path.join(d, f"expected_{vname}.txt"), "wb") as f:
                f.write(out)
            # the reference CLI itself (`kallisto quant -t 1 --plaintext`): abundance.tsv, bs_abundance_*.tsv, run_info.json
            cli = [a.replace("--fr", "--fr-stranded").replace("--rf", "--rf-stranded") for a in extra]
            cli = ["-b" if a == "--boot" else a for a in cli]
            od = os.path.join(tmp, "q_" + vname)
            subprocess.run([KALLISTO, "quant", "-i", idx, "-o", od, "-t", "1", "--plaintext", *cli, *vfiles], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            cd = os.path.join(d, "cli_" + vname)
            os.makedirs(cd, exist_ok=True)
            for fn in sorted(os.listdir(od)):
                if fn.endswith(".tsv"):
                    shutil.copy(os.path.join(od, fn), os.path.join(cd, fn))
            # abundance.h5 of a USE_HDF5 build of the reference (oracle/_ref/kallisto_h5, `make -C oracle ref_h5`), as h5dump text
            if (name, vname) in H5_VARIANTS and os.path.exists(KALLISTO_H5):
                oh = os.path.join(tmp, "h_" + vname)
                env = dict(os.environ, LD_PRELOAD="/usr/lib/x86_64-linux-gnu/libstdc++.so.6")  # the conda tree ships an older libstdc++
                subprocess.run([KALLISTO_H5, "quant", "-i", idx, "-o", oh, "-t", "1", *cli, *vfiles], check=True, env=env,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                dump = subprocess.run([H5DUMP, "-p", "-m", "%.12g", os.path.join(oh, "abundance.h5")], check=True,
                                      stdout=subprocess.PIPE).stdout.decode()
                dump = h5dump_tools.collapse_constant_bias(dump.replace(os.path.join(oh, "abundance.h5"), "abundance.h5"))
                with open(os.path.join(cd, "abundance.h5.dump"), "w") as f:
                    f.write(dump)
            info = json.load(open(os.path.join(od, "run_info.json")))
            for key in ("start_time", "call"):
                info.pop(key, None)
            json.dump(info, open(os.path.join(cd, "run_info.json"), "w"), indent=1)
    with open(os.path.join(d, "case.json"), "w") as f:
        json.dump({"name": name, "k": k, "paired": reads2 is not None, "n": len(reads1), "variants": variants,
                   "note": note, "reference": "pachterlab/kallisto v0.51.1 via oracle/_ref/dump_ec (unmodified sources)"},
                  f, indent=1)
    print(name, "index bytes", os.path.getsize(idx))


def read_fastq_gz(path):
    out = []
    with gzip.open(path, "rb") as f:
        for i, line in enumerate(f):
            if i % 4 == 1:
                out.append(line.rstrip(b"\r\n"))
    return out


def main():
    if not (os.path.exists(KALLISTO) and os.path.exists(DUMP)):
        sys.exit("oracle/_ref is not built: run `make -C oracle ref` in the build container")
    tmp = tempfile.mkdtemp()
    # 1. the reference's own test data (BASELINE config #1): 14 transcripts, 10 000 PE-50 pairs
    T = "/root/reference/test/"
    fa = os.path.join(tmp, "t.fa")
    with gzip.open(T + "transcripts.fasta.gz", "rb") as fi, open(fa, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    make_case("ref_test_pe", fa, read_fastq_gz(T + "reads_1.fastq.gz"), read_fastq_gz(T + "reads_2.fastq.gz"),
              {"pe": [], "pe_boot": ["--boot", "3", "--seed", "42"], "pe_l200": ["-l", "200", "-s", "20"],
               "pe_rf": ["--rf"], "pe_fr": ["--fr"], "pe_nojump": ["--no-jump"], "pe_nojump_fr": ["--no-jump", "--fr"],
               "pe_union": ["--union"]},
              note="test/transcripts.fasta.gz + test/reads_{1,2}.fastq.gz of the reference repository")
    # 2. yeast-like (config #2 shape, scaled down): single-end with errors and N's
    seqs = synth.yeast_like(n_tr=300, seed=1)
    fa = os.path.join(tmp, "y.fa")
    synth.write_fasta(fa, seqs)
    r1, _ = synth.simulate_reads(seqs, 6000, 100, paired=False, frag_mean=200, frag_sd=20, err=0.01, n_frac=0.02, seed=21)
    se = ["--single", "-l", "200", "-s", "20"]
    make_case("yeast_se", fa, list(r1), None,
              {"se": se, "se_overhang": se + ["--single-overhang"], "se_fr": se + ["--fr"], "se_rf": se + ["--rf"],
               "se_nojump": se + ["--no-jump"], "se_nojump_rf": se + ["--no-jump", "--rf"]},
              note="synth.yeast_like(300, seed=1); simulate_reads(6000 SE-100, err 1%, 2% reads with an N, seed=21)")
    # 3. human-like (config #3 shape, scaled down): isoform families -> mosaic ECs; paired-end
    seqs = synth.human_like(n_genes=60, seed=2)
    fa = os.path.join(tmp, "h.fa")
    synth.write_fasta(fa, seqs)
    r1, r2 = synth.simulate_reads(seqs, 5000, 100, paired=True, err=0.005, n_frac=0.01, seed=22)
    # a few ragged / degenerate reads: shorter than k, all-N, lower case, trailing N runs
    r1 = [bytes(x) for x in r1]
    r2 = [bytes(x) for x in r2]
    r1[10] = r1[10][:20]; r2[11] = r2[11][:30]; r1[12] = b"N" * 100; r2[13] = r2[13].lower()
    r1[14] = r1[14][:60] + b"N" * 40; r2[15] = b"N" * 35 + r2[15][35:]; r1[16] = r1[16][:31]; r2[17] = r2[17][:75]
    make_case("human_pe", fa, r1, r2, {"pe": [], "pe_boot": ["--boot", "2", "--seed", "7"], "pe_l180": ["-l", "180", "-s", "25"],
                                        "pe_rf": ["--rf"], "pe_nojump": ["--no-jump"], "pe_union": ["--union"]},
              note="synth.human_like(60 genes, seed=2); simulate_reads(5000 PE-100, seed=22) + ragged/degenerate reads")
    # 4. small k, very short reads (the shape of func_tests/runtests.sh): k=7
    rng = np.random.default_rng(5)
    seqs = [synth._ACGT[rng.integers(0, 4, int(l))] for l in rng.integers(60, 200, 12)]
    fa = os.path.join(tmp, "s.fa")
    synth.write_fasta(fa, seqs)
    r1, _ = synth.simulate_reads(seqs, 3000, 24, paired=False, frag_mean=40, frag_sd=5, err=0.01, n_frac=0.01, seed=23)
    se = ["--single", "-l", "40", "-s", "5"]
    make_case("tiny_k7_se", fa, list(r1), None, {"se": se, "se_overhang": se + ["--single-overhang"]}, k=7,
              note="12 random transcripts of 60-200 bp, k=7, 3000 SE-24 reads (func_tests-like)")
    # 5. an index with a D-list (`kallisto index --d-list`, what kb-python builds): "genomic" sequences that share a stretch
    #    with a transcript and then diverge yield distinguishing flanking k-mers; reads drawn from them (pre-mRNA-like) cross
    #    those k-mers and must be discarded (KmerIndex.cpp:1818-1826,1928-1939)
    seqs = synth.human_like(n_genes=60, seed=2)
    fa = os.path.join(tmp, "dt.fa")
    synth.write_fasta(fa, seqs)
    rng = np.random.default_rng(9)
    genomic = []
    for i in rng.choice(len(seqs), 80, replace=False):
        s = seqs[i]
        if len(s) < 400:
            continue
        a = int(rng.integers(0, len(s) - 300)); ln = int(rng.integers(150, 300))
        genomic.append(np.concatenate([synth._ACGT[rng.integers(0, 4, 200)], s[a:a + ln], synth._ACGT[rng.integers(0, 4, 200)]]))
    n_shared = len(genomic)
    for _ in range(10):
        genomic.append(synth._ACGT[rng.integers(0, 4, 1000)])
    gfa = os.path.join(tmp, "dg.fa")
    synth.write_fasta(gfa, genomic)
    r1a, r2a = synth.simulate_reads(seqs, 3000, 100, paired=True, err=0.005, n_frac=0.01, seed=31)
    r1b, r2b = synth.simulate_reads(genomic[:n_shared], 1500, 100, paired=True, frag_mean=180, frag_sd=20, err=0.002, n_frac=0.0, seed=32)
    r1 = [bytes(x) for x in r1a] + [bytes(x) for x in r1b]
    r2 = [bytes(x) for x in r2a] + [bytes(x) for x in r2b]
    perm = np.random.default_rng(3).permutation(len(r1))
    r1 = [r1[i] for i in perm]; r2 = [r2[i] for i in perm]
    se = ["--single", "-l", "180", "-s", "20"]
    make_case("dlist_pe", fa, r1, r2, {"pe": [], "pe_fr": ["--fr"], "se": se, "se_rf": se + ["--rf"], "pe_nojump": ["--no-jump"],
                                    "se_nojump": se + ["--no-jump"], "pe_union": ["--union"]}, index_args=["--d-list=" + gfa],
              note="synth.human_like(60 genes, seed=2) + a D-list of 80 transcript-fragment-in-random-flanks sequences and 10 random "
                   "ones; 3000 PE-100 pairs from the transcripts + 1500 from the D-list sequences, shuffled")
    # 6. reads that switch between close paralogs every 15-45 bases: here the jumps of match() skip k-mers that a full scan
    #    sees, so `--no-jump` changes the outcome (on the cases above it does not)
    rng = np.random.default_rng(11)
    fam = []
    for _ in range(25):
        base = synth._ACGT[rng.integers(0, 4, int(rng.integers(500, 1200)))]
        fam.append([base])
        for _ in range(2):
            m = base.copy()
            sites = rng.random(len(m)) < 0.02
            m[sites] = synth._ACGT[(np.searchsorted(synth._ACGT, m[sites]) + rng.integers(1, 4, int(sites.sum()))) % 4]
            fam[-1].append(m)
    seqs = [m for f in fam for m in f]
    fa = os.path.join(tmp, "m.fa")
    synth.write_fasta(fa, seqs)
    comp = bytes.maketrans(b"ACGT", b"TGCA")

    def mosaic(f, a, ln):
        out = np.empty(ln, np.uint8); i = 0
        while i < ln:
            seg = int(rng.integers(15, 46)); mem = f[int(rng.integers(0, 3))]
            out[i:i + seg] = mem[a + i:a + min(i + seg, ln)]; i += seg
        return bytes(out)
    r1, r2 = [], []
    for _ in range(4000):
        f = fam[int(rng.integers(0, len(fam)))]
        fl = int(rng.integers(150, 260)); a = int(rng.integers(0, len(f[0]) - fl))
        m1 = mosaic(f, a, 75); m2 = mosaic(f, a + fl - 75, 75).translate(comp)[::-1]
        if rng.random() < 0.5:
            r1.append(m1); r2.append(m2)
        else:
            r1.append(m2); r2.append(m1)
    se = ["--single", "-l", "200", "-s", "25"]
    make_case("mosaic_pe", fa, r1, r2, {"pe": [], "pe_nojump": ["--no-jump"], "se": se, "se_nojump": se + ["--no-jump"],
                                         "pe_nojump_rf": ["--no-jump", "--rf"], "pe_rf": ["--rf"], "pe_union": ["--union"],
                                         "se_union_overhang": se + ["--single-overhang", "--union"], "pe_union_fr": ["--union", "--fr"]},
              note="25 families of 3 paralogs (2 % divergence); 4000 PE-75 pairs whose mates switch between the paralogs every 15-45 "
                   "bases -- the case where --no-jump differs from the default.  (--single --union without --single-overhang is "
                   "not a variant: the reference itself dies there with 'Index not present in SparseVector' -- the union holds "
                   "transcripts that the first mapping k-mer's set does not, and findPosition looks them up in it.)")
    # 7. the structure a real transcriptome has (synth.human_stress): repeat-family copies in terminal exons, paralog families, poly-A
    #    tails -- equivalence classes of hundreds of transcripts, reads with more than eight distinct (unitig, set) classes -- and reads
    #    the benches of rounds 1-4 never saw: 12 % off-transcriptome pairs (background with copies of the same repeats: partial k-mer
    #    matches), a 3' quality tail of substitution errors
    seqs, bg = synth.human_stress(n_genes=150, seed=7, n_repeat_families=2, paralog_members=30, polya_frac=0.05, background_mbp=0.2)
    fa = os.path.join(tmp, "st.fa")
    synth.write_fasta(fa, seqs)
    r1, r2 = synth.simulate_reads(seqs, 6000, 100, paired=True, err=0.002, n_frac=0.005, seed=41, background=bg, off_frac=0.12, tail_err=0.05)
    se = ["--single", "-l", "200", "-s", "25"]
    make_case("stress_pe", fa, [bytes(x) for x in r1], [bytes(x) for x in r2],
              {"pe": [], "pe_union": ["--union"], "pe_rf": ["--rf"], "se": se, "pe_nojump": ["--no-jump"], "pe_boot": ["--boot", "2", "--seed", "11"]},
              note="synth.human_stress(150 genes, seed=7: 2 repeat families in 30 % of the genes' terminal exons, 3 x 30 paralog genes, 5 % poly-A tails); "
                   "simulate_reads(6000 PE-100, seed=41, 12 % off-transcriptome pairs from a background with repeat copies, 3' quality tail up to 5 %)")
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
