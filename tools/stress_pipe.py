#!/usr/bin/env python3
"""Stress of the batch pipeline over several cores on ONE GPU (what `basal -G 0,0,0` builds), in one process: the text of a golden fixture goes
through a pipe over `ncores` cores in small batches, a submitter thread and a collector thread as in the command line, `iters` times, and every
run's SAM bytes must be the golden ones.  On a difference the batch, the record and both lines are printed (and appended to --log).

  python tools/stress_pipe.py tdel_pipeline 3 20000 2000 [--log profiles/r04_stress_multi.log]
"""
import argparse
import gzip
import os
import sys
import threading
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cut_batches(text, nbytes, lines_per_record):
    """Consecutive whole records, each batch as many as fit into nbytes (at least one) -- the command line's cut (basal_main.cpp, run_se)."""
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    recs = [b"\n".join(lines[i:i + lines_per_record]) + b"\n" for i in range(0, len(lines), lines_per_record)]
    out, cur, size = [], [], 0
    for r in recs:
        if cur and size + len(r) > nbytes:
            out.append(b"".join(cur))
            cur, size = [], 0
        cur.append(r)
        size += len(r)
    if cur:
        out.append(b"".join(cur))
    return out


def run_once(B, pipe, batches, fmt):
    """One pass of all batches through the pipe: submitter and collector on their own threads. Returns the list of per-batch outputs."""
    got, err = [], []
    sem = threading.Semaphore(0)

    def collector():
        for _ in batches:
            sem.acquire()
            rc, data, _st = pipe.collect()
            if rc:
                err.append((rc, data))
                return
            got.append(data)
    t = threading.Thread(target=collector)
    t.start()
    try:
        for b in batches:
            pipe.submit_text(b, fmt=fmt)  # (acquire blocks while every slot is in flight)
            sem.release()
    finally:
        t.join()
    if err:
        raise RuntimeError("pipe_collect failed: %r" % (err[0],))
    return got


def stress(name, ncores, nbytes, iters, log=None, depth=2, verbose=True):
    import basal_amd as B
    import harness as H
    fa, fq, _, _ = H.fixture_paths(name)
    flags = H.MANIFEST[name]["flags"]
    p = B.Params(H.rule_of(flags), flags)
    ref = B.Reference(p, fasta_path=fa)
    ref.build_index(4)
    cores = [B.Core(p) for _ in range(ncores)]
    for c in cores:
        c.upload(ref)
        c.set_contig_names(ref.names())
    text = gzip.open(fq, "rb").read() if fq.endswith(".gz") else open(fq, "rb").read()
    fasta = text[:1] == b">"
    batches = cut_batches(text, nbytes, 2 if fasta else 4)
    gold = H.golden_sam(name)
    body = "".join(l + "\n" for l in gold.split("\n") if l and not l.startswith("@")).encode()
    bad = 0
    t0 = time.time()
    msgs = []
    for it in range(iters):
        pipe = B.Pipe(cores if ncores > 1 else cores[0], depth=depth, max_reads=4096, max_bytes=1 << 20)
        pipe.set_read_range(0)
        try:
            got = run_once(B, pipe, batches, B.FMT_FASTA if fasta else B.FMT_FASTQ)
        finally:
            pipe.close()
        out = b"".join(got)
        if out != body:
            bad += 1
            g, e = out.split(b"\n"), body.split(b"\n")
            # which batch holds the first differing record
            first = next((k for k, (a, b_) in enumerate(zip(g, e)) if a != b_), min(len(g), len(e)))
            nrec, bno = 0, -1
            for k, o in enumerate(got):
                nrec += o.count(b"\n")
                if first < nrec:
                    bno = k
                    break
            m = "iteration %d: output differs (%d vs %d lines); first differing record %d in batch %d of %d\n   got: %s\n   exp: %s" % (
                it, len(g), len(e), first, bno, len(batches), g[first][:300].decode("latin1") if first < len(g) else "<none>",
                e[first][:300].decode("latin1") if first < len(e) else "<none>")
            msgs.append(m)
            if verbose:
                print(m, flush=True)
        if verbose and (it + 1) % 200 == 0:
            print("[stress] %s x%d cores, %d-byte batches (%d per pass): %d of %d passes bad, %.0f s" % (name, ncores, nbytes, len(batches), bad, it + 1, time.time() - t0), flush=True)
    summary = "%s: %d cores on one GPU, %d-byte batches (%d per pass), depth %d: %d bad of %d passes in %.0f s" % (name, ncores, nbytes, len(batches), depth, bad, iters, time.time() - t0)
    if log:
        with open(log, "a") as f:
            for m in msgs:
                f.write(m + "\n")
            f.write(summary + "\n")
    if verbose:
        print(summary, flush=True)
    return bad, msgs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("ncores", type=int)
    ap.add_argument("nbytes", type=int)
    ap.add_argument("iters", type=int)
    ap.add_argument("--log", default=None)
    ap.add_argument("--depth", type=int, default=2)
    a = ap.parse_args()
    bad, _ = stress(a.name, a.ncores, a.nbytes, a.iters, a.log, a.depth)
    sys.exit(1 if bad else 0)
