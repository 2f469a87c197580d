#!/usr/bin/env python3
"""End-to-end rate of the `basal` command line on the GPU box: FASTQ file in, SAM file out.

  python3 tools/bench_cli.py [--scale 1.0] [--reads 10000000] [--rule C:T] [--flags "..."] [--out /dev/shm/out.sam]

Generates a genome FASTA (tools/synth_gpu.py, hg38-sized at --scale 1.0) and a FASTQ file, runs basal_amd/bin/basal on them and
prints one JSON line with the wall-clock phases the binary reports.  Bench tooling; not part of the product path."""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--rule", default="C:T")
    ap.add_argument("--flags", default="-S 1")
    ap.add_argument("--dir", default="/dev/shm/basal_bench")
    ap.add_argument("--out", default=None)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--runs", type=int, default=2)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--realistic", action="store_true", help="the hg38-like repeat landscape (bench.py's default genome) instead of the uniform stand-in")
    ap.add_argument("--pairs", type=int, default=0, help="paired-end: this many read pairs (two FASTQ files, basal -a/-b) instead of --reads single reads")
    a = ap.parse_args()
    import numpy as np
    import torch
    import basal_amd as B
    import synth_files
    import synth_gpu
    os.makedirs(a.dir, exist_ok=True)
    fa, fq = os.path.join(a.dir, "g.fa"), os.path.join(a.dir, "r.fq")
    out = a.out or os.path.join(a.dir, "out.sam")
    dev = torch.device("cuda", 0)
    p = B.Params(a.rule, ["-M", a.rule])
    t0 = time.time()
    G = synth_gpu.make_genome(p, dev, scale=a.scale, seed=1, repeat_copies=40000, realistic=a.realistic)
    synth_files.write_fasta(fa, G)
    frm = "ACGT".index(a.rule[0])
    tos = [t for t in a.rule[2:] if t in "ACGT"]
    to = "ACGT".index(tos[0]) if tos else frm
    fq2 = os.path.join(a.dir, "r2.fq")
    if a.pairs:
        a.reads = 2 * a.pairs
        with open(fq, "wb") as f1, open(fq2, "wb") as f2:
            per = 1_000_000
            for b0 in range(0, a.pairs, per):
                nb = min(per, a.pairs - b0)
                m1, m2 = synth_gpu.make_pairs(G, nb, dev, read_len=a.read_len, seed=200 + b0 // per, conv_from=frm, conv_to=to)
                f1.write(synth_files.fastq_bytes(m1.cpu().numpy().reshape(nb, a.read_len), np.full(nb, a.read_len), None, first=b0))
                f2.write(synth_files.fastq_bytes(m2.cpu().numpy().reshape(nb, a.read_len), np.full(nb, a.read_len), None, first=b0))
    with open(fq, "ab" if a.pairs else "wb") as f:
        per = 2_000_000 if not a.pairs else 1 << 62
        for b0 in range(0, a.reads if not a.pairs else 0, per):
            nb = min(per, a.reads - b0)
            bases, _, _, _ = synth_gpu.make_reads(G, nb, dev, read_len=a.read_len, seed=100 + b0 // per, conv_from=frm, conv_to=to, p_conv=0.95 if tos else 0.0)
            s = bases.cpu().numpy().reshape(nb, a.read_len)
            f.write(synth_files.fastq_bytes(s, np.full(nb, a.read_len), None, first=b0))
    del G
    torch.cuda.empty_cache()
    t_gen = time.time() - t0
    best = None
    for run in range(a.runs):
        cmd = [os.path.join(ROOT, "basal_amd", "bin", "basal"), "-a", fq] + (["-b", fq2] if a.pairs else []) + ["-d", fa, "-M", a.rule] + a.flags.split() + ["-p", str(a.threads), "-o", out]
        t1 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True)
        wall = time.time() - t1
        if r.returncode != 0:
            raise SystemExit("basal failed:\n" + r.stderr)
        if a.pairs:
            m = re.search(r"align phase ([0-9.]+) s = ([0-9.]+) Mpairs/s; GPU batches ([0-9.]+) s", r.stderr)
            d = {"wall_s": round(wall, 2), "align_phase_s": float(m.group(1)), "mpairs_per_s": float(m.group(2)), "mreads_per_s": 2 * float(m.group(2)), "gpu_batches_s": float(m.group(3))}
            print("[bench_cli] run %d: %s" % (run, d), file=sys.stderr)
            if best is None or d["mreads_per_s"] > best["mreads_per_s"]:
                best = d
            last_log = r.stderr
            continue
        m = re.search(r"align phase ([0-9.]+) s = ([0-9.]+) Mreads/s; GPU stage sums: H2D ([0-9.]+), read prep ([0-9.]+), align ([0-9.]+), SAM ([0-9.]+), D2H ([0-9.]+)", r.stderr)
        d = {"wall_s": round(wall, 2), "align_phase_s": float(m.group(1)), "mreads_per_s": float(m.group(2)), "gpu_h2d_s": float(m.group(3)), "gpu_prep_s": float(m.group(4)),
             "gpu_align_s": float(m.group(5)), "gpu_sam_s": float(m.group(6)), "gpu_d2h_s": float(m.group(7))}
        print("[bench_cli] run %d: %s" % (run, d), file=sys.stderr)
        if best is None or d["mreads_per_s"] > best["mreads_per_s"]:
            best = d
        last_log = r.stderr
    res = {"what": "basal CLI end to end (FASTQ file -> SAM file)", "reads": a.reads, "read_len": a.read_len, "genome_scale": a.scale, "rule": a.rule, "flags": a.flags,
           "threads": a.threads, "fastq_bytes": os.path.getsize(fq), "sam_bytes": os.path.getsize(out), "out": out, "gen_s": round(t_gen, 1), "best": best,
           "log_tail": last_log.strip().splitlines()[-4:]}
    print(json.dumps(res))
    if not a.keep:
        for f in (fa, fq, fq2, out):
            try:
                os.remove(f)
            except OSError:
                pass


if __name__ == "__main__":
    main()
