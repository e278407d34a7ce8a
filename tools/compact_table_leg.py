#!/usr/bin/env python3
"""A side leg of bench.py, in a process of its own: the steps of the headline on the COMPACT layout of the k-mer table
(KAMD_TABLE_LAYOUT=compact, kallisto_amd/csrc/kamd_core.h: four exact 16-byte slots per 64-byte line by quotienting) at a few load factors.

bench.py writes the 2-bit packed reads of its run to a directory under /dev/shm (words.i32, lens.i16) and starts this script; the last line
of stdout is a JSON list, one entry per load factor (table size, ms per full quant, kernel A's time, bucket lines per item), and
<dir>/result_<i>.npz holds what the parent compares with its own (wide) result: counts, fragment lengths, est_counts, EM rounds.
Not a benchmark of its own: no warm-up policy, no roofline -- `python bench.py --table-layout compact` is the full line on that layout."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--index", required=True)
    ap.add_argument("--dir", required=True)
    ap.add_argument("--items", type=int, required=True)
    ap.add_argument("--read-len", type=int, required=True)
    ap.add_argument("--paired", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--loads", default="0.6,0.5")
    a = ap.parse_args()

    import torch
    import kallisto_amd as ka

    dev = torch.device("cuda", a.device)
    torch.cuda.set_device(a.device)
    paired = bool(a.paired)
    per = 2 if paired else 1
    rec = ka.packed_record_words(a.read_len)
    n, L = a.items, a.read_len
    words = torch.from_numpy(np.fromfile(os.path.join(a.dir, "words.i32"), np.int32)).to(dev)
    lens = torch.from_numpy(np.fromfile(os.path.join(a.dir, "lens.i16"), np.int16)).to(dev)
    assert words.numel() == n * per * rec and lens.numel() == n * per
    opts = ka.QuantOpts(1, 0.0, 0.0, 0, 0) if paired else ka.QuantOpts(0, 200.0, 20.0, 0, 0)
    out = []
    for i, tok in enumerate(a.loads.split(",")):   # a load factor of the compact table, or "wide"
        wide = tok.strip() == "wide"
        load = 0.5 if wide else float(tok)
        entry = {"layout": "wide" if wide else "compact", "load_asked": load}
        try:
            os.environ["KAMD_TABLE_LAYOUT"] = "wide" if wide else "compact"
            os.environ["KAMD_TABLE_LOAD"] = str(load)
            t0 = time.time()
            index = ka.Index(a.index)
            entry["index_flatten_s"] = round(time.time() - t0, 2)
            v = index.view
            entry.update({"slots_per_line": int(v.slots_per_bucket), "table_bytes": int((v.n_buckets + v.pad_buckets) * 64),
                          "load": round(index.num_kmers / float(v.slots_per_bucket * v.n_buckets), 3), "tag_bits": int(v.tag_w)})
            if v.table_layout != (0 if wide else 1):
                raise RuntimeError("the loader did not build the layout asked for")
            ctx = ka.Context(a.device)
            ctx.upload(index)

            def step():
                ctx.reset()
                return ka.quant(ctx, opts, [(words, lens, n, L)], download_ecs=False)
            for _ in range(max(a.warmup, 1)):
                res = step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            profs = []
            for _ in range(a.steps):
                res = step()
                profs.append(ctx.profile())
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            st = res.stats
            entry.update({"ms_per_step": round(el / a.steps * 1e3, 3), "value": round(n * a.steps / el / 1e6, 4),
                          "unit": "M read pairs/s" if paired else "M reads/s",
                          "kernel_a_ms": round(float(np.mean([q["align_kernel_ms"] for q in profs])), 3),
                          "em_ms": round(float(np.mean([q["em_ms"] for q in profs])), 3),
                          "bucket_reads_per_item": round(st["n_bucket_reads"] / max(st["n_processed"], 1), 3),
                          "text_hits_per_item": round(st["n_text_hits"] / max(st["n_processed"], 1), 3),
                          "probes_per_item": round(st["n_probes"] / max(st["n_processed"], 1), 3)})
            np.savez(os.path.join(a.dir, f"result_{i}.npz"), n_pseudoaligned=res.n_pseudoaligned, n_unique=res.n_unique, flens=res.flens,
                     est_counts=res.est_counts, em_rounds=res.em_rounds)
            ctx.close()
            index.close()
        except Exception as e:   # noqa: BLE001
            entry["error"] = str(e)[:300]
        out.append(entry)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
