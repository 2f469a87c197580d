#!/usr/bin/env python3
"""The like-for-like CPU number: the UNMODIFIED reference binary (oracle/_ref/basal, built from /root/reference by oracle/Makefile.ref and shipped
with the snapshot) on the bench line's own workload -- bench.py's hg38-sized stand-in genome (3.09 Gbp, hg38-like repeat landscape) written to
FASTA, its own index at the default -s 16, and reads made the way bench.py makes them -- timed on this box's host cores; and the product's
command line on the same two files, whose SAM must be the reference's (as a set of records: the reference's -p N writes whole batches in
any order).  bench.py's cpu_baseline stays the density-equivalent 50 Mbp proxy (a default run must finish in minutes); this job takes a
quarter of an hour and its result is recorded in BASELINE.md.

  gpurun --timeout 1200 -- 'python3 tools/ref_like_for_like.py --reads 400000 > gpurun_out/ref_like_for_like.json'
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def log(*a):
    print("[like-for-like]", *a, file=sys.stderr, flush=True)


def records_digest(path):
    """(md5 of the sorted alignment records, their number): the reference's worker threads write whole batches in any order."""
    recs = sorted(l for l in open(path, "rb") if not l.startswith(b"@"))
    h = hashlib.md5()
    for l in recs:
        h.update(l)
    return h.hexdigest(), len(recs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=400_000)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--genome", default="realistic", choices=["realistic", "uniform"])
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    import numpy as np
    import torch
    import basal_amd as B
    import synth_files
    import synth_gpu
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "basal")
    gpu_bin = os.path.join(ROOT, "basal_amd", "bin", "basal")
    threads = a.threads or min(16, os.cpu_count() or 1)
    dev = torch.device("cuda", 0)
    p = B.Params("C:T", ["-M", "C:T", "-S", "1"])
    d = tempfile.mkdtemp(prefix="basal_l4l_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    out = {"workload": "config 2: %d synthetic 100 bp SE reads, -M C:T -g 0 -S 1 -s 16, bench.py's hg38-sized stand-in genome (%s, scale %.2f) as a FASTA file" % (a.reads, a.genome, a.scale),
           "threads": threads}
    try:
        t0 = time.time()
        G = synth_gpu.make_genome(p, dev, scale=a.scale, seed=1, repeat_copies=40000, realistic=a.genome == "realistic")
        fa, fq = os.path.join(d, "g.fa"), os.path.join(d, "r.fq")
        synth_files.write_fasta(fa, G)
        out["genome_bp"] = int(sum(G.sizes))
        with open(fq, "wb") as f:
            for b0 in range(0, a.reads, 400_000):
                nb = min(400_000, a.reads - b0)
                # (bench.py's read recipe for config 2: C -> T with p 0.95, 1 % substitutions, both strands)
                bases, _, _, _ = synth_gpu.make_reads(G, nb, dev, read_len=100, seed=1005 + b0 // 400_000, conv_from=1, conv_to=3, p_conv=0.95)
                f.write(synth_files.fastq_bytes(bases.cpu().numpy().reshape(nb, 100), np.full(nb, 100), None, first=b0))
        del G
        torch.cuda.empty_cache()
        log("files written in %.0f s: %s (%.2f GB), %s" % (time.time() - t0, fa, os.path.getsize(fa) / 1e9, fq))
        flags = ["-M", "C:T", "-S", "1"]

        def wall(cmd):
            t = time.perf_counter()
            r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
            if r.returncode != 0:
                raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stderr[-500:]))
            return time.perf_counter() - t
        ref_sam, gpu_sam = os.path.join(d, "ref.sam"), os.path.join(d, "gpu.sam")
        t_idle = wall([ref_bin, "-a", fq, "-d", fa] + flags + ["-p", str(threads), "-E", "0", "-o", os.path.join(d, "idle.sam")])
        log("reference, load + index build only (-E 0): %.1f s" % t_idle)
        t_full = wall([ref_bin, "-a", fq, "-d", fa] + flags + ["-p", str(threads), "-o", ref_sam])
        log("reference, full run: %.1f s" % t_full)
        t_gpu = wall([gpu_bin, "-a", fq, "-d", fa] + flags + ["-p", str(threads), "-o", gpu_sam])
        secs = max(t_full - t_idle, 1e-3)
        dr, dg = records_digest(ref_sam), records_digest(gpu_sam)
        out.update({"reference_load_and_index_s": round(t_idle, 1), "reference_full_s": round(t_full, 1), "reference_align_s": round(secs, 2),
                    "reference_mreads_per_s": a.reads / secs / 1e6, "cpu_seconds_of_alignment": round(secs * threads, 1),
                    "product_cli_whole_run_s": round(t_gpu, 1), "sam_records": dr[1], "sam_identical_as_a_set": dr == dg,
                    "sam_md5_sorted_records": dr[0]})
        print(json.dumps(out))
        if dr != dg:
            log("SAM DIFFERS: reference %r product %r" % (dr, dg))
            sys.exit(1)
    finally:
        if not a.keep:
            shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
