"""Lab: the front-end from FASTQ files at a size where steady state shows (GPU box).  python scratch/e2e_lab.py [pairs] [genes]
Writes FASTQ (plain, BGZF, gzip subset) under /tmp and runs kallisto_amd_quant --verbose in several settings, printing its stage lines."""
import os, subprocess, sys, time, struct, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B

def bgzf_block(chunk):
    co = zlib.compressobj(1, zlib.DEFLATED, -15)
    comp = co.compress(chunk) + co.flush()
    return (b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(comp) + 8 - 1)
            + comp + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))

def _bgzf_part(args):
    src, a, b = args
    with open(src, "rb") as f:
        f.seek(a); data = f.read(b - a)
    return b"".join(bgzf_block(data[i:i + 65280]) for i in range(0, len(data), 65280))

def make_bgzf(src, dst, procs=48):
    import multiprocessing as mp
    size = os.path.getsize(src)
    step = 65280 * 256
    parts = [(src, a, min(a + step, size)) for a in range(0, size, step)]
    with mp.Pool(procs) as pool, open(dst, "wb") as fo:
        for blob in pool.imap(_bgzf_part, parts):
            fo.write(blob)
        fo.write(bgzf_block(b""))

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
    genes = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    import torch
    from kallisto_amd.synth_gpu import ReadSimulator
    cat, tlens, idx = B.prepare_workload("human", genes, True)
    dev = torch.device("cuda", 0)
    sim = ReadSimulator(cat, tlens, dev, seed=1000, read_len=100)
    tmp = "/tmp/e2e_lab"; os.makedirs(tmp, exist_ok=True)
    f1, f2 = tmp + "/r_1.fq", tmp + "/r_2.fq"
    t0 = time.time()
    with open(f1, "wb") as a, open(f2, "wb") as b:
        for s in range(0, n, 4_000_000):
            m = min(4_000_000, n - s)
            r1, r2 = sim.draw(m)
            for fo, r in ((a, r1), (b, r2)):
                p = tmp + "/part.fq"
                B.write_fastq_fast(p, r.cpu().numpy())
                fo.write(open(p, "rb").read())
    print(f"wrote 2 x {os.path.getsize(f1)/1e9:.2f} GB in {time.time()-t0:.1f}s", flush=True)
    del sim; torch.cuda.empty_cache()
    if len(sys.argv) > 3 and sys.argv[3] == "files-only":
        return
    exe = os.path.join(B.ROOT, "kallisto_amd", "kallisto_amd_quant")
    def run(tag, files, cnt, threads=64, env=None, extra=()):
        e = dict(os.environ); e.update(env or {})
        t0 = time.time()
        p = subprocess.run([exe, "quant", "-i", idx, "-o", tmp + "/out", "-t", str(threads), "--plaintext", "--verbose", *extra, *files], env=e,
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        wall = time.time() - t0
        lines = [l for l in p.stderr.decode(errors="replace").splitlines() if l.startswith("[timing]") or "device parser" in l or "Error" in l]
        tm = {}
        for l in lines:
            if l.startswith("[timing] index file read"): tm["idx"] = float(l.split()[-2])
            if l.startswith("[timing] reads parsed"): tm["reads"] = float(l.split()[-2])
        rate = cnt / max(tm.get("reads", wall) - tm.get("idx", 0), 1e-9) / 1e6
        print(f"== {tag}: rc={p.returncode} wall={wall:.2f}s input->ECs {rate:.1f} M pairs/s", flush=True)
        for l in lines: print("   ", l[:400], flush=True)
    run("plain t64 (capped to the cgroup)", [f1, f2], n)
    run("plain t64 again", [f1, f2], n)
    run("plain KAMD_CPUS=12", [f1, f2], n, env={"KAMD_CPUS": "12"})
    run("plain KAMD_CPUS=20", [f1, f2], n, env={"KAMD_CPUS": "20"})
    run("plain unit 64MB", [f1, f2], n, env={"KAMD_FQ_UNIT_MB": "64", "KAMD_FQ_BUFS": "12"})
    run("plain unit 16MB", [f1, f2], n, env={"KAMD_FQ_UNIT_MB": "16", "KAMD_FQ_BUFS": "40"})
    run("plain ring 512MB", [f1, f2], n, env={"KAMD_FQ_RING_MB": "512"})
    run("plain host-parse", [f1, f2], n, env={"KAMD_HOST_PARSE": "1"})
    t0 = time.time(); make_bgzf(f1, tmp + "/b_1.fq.gz"); make_bgzf(f2, tmp + "/b_2.fq.gz"); print(f"bgzf written in {time.time()-t0:.1f}s", flush=True)
    run("bgzf t64", [tmp + "/b_1.fq.gz", tmp + "/b_2.fq.gz"], n)
    run("bgzf KAMD_CPUS=20", [tmp + "/b_1.fq.gz", tmp + "/b_2.fq.gz"], n, env={"KAMD_CPUS": "20"})
    run("bgzf t64 zlib", [tmp + "/b_1.fq.gz", tmp + "/b_2.fq.gz"], n, env={"KAMD_NO_LIBDEFLATE": "1"})
    ng = min(n, 2_000_000); per = os.path.getsize(f1) // n
    for src, dst in ((f1, tmp + "/g_1.fq"), (f2, tmp + "/g_2.fq")):
        with open(src, "rb") as fi, open(dst, "wb") as fo: fo.write(fi.read(per * ng))
    t0 = time.time()
    ps = [subprocess.Popen(["gzip", "-1", "-f", tmp + f"/g_{i}.fq"]) for i in (1, 2)]
    [p.wait() for p in ps]; print(f"gzip -1 of 2 x {per*ng/1e6:.0f} MB in {time.time()-t0:.1f}s", flush=True)
    run("gzip", [tmp + "/g_1.fq.gz", tmp + "/g_2.fq.gz"], ng)

if __name__ == "__main__":
    main()
