#!/usr/bin/env python3
"""Many tiny contigs (a transcriptome-shaped reference) + reads, from a counter-based generator.

Used for the contig-count edge fixtures: > 4 096 contigs (three levels of the 64-ary contig search) and > 131 071
contigs (the reference keeps the contig number in an 18-bit field, param.h:35-42, so `chr = 2*contig + strand`
wraps; align.cpp:319-346).  The 140 000-contig FASTA would be several MB compressed, so it is NOT committed: it
is regenerated from (contigs, seed) wherever it is needed, and its SHA-256 sits in the manifest next to the golden
SAM the reference printed for it.  For that the bytes must never change, so nothing here uses numpy's Generator
(whose algorithms may change between versions): every value is splitmix64(counter), plain uint64 arithmetic.

Test/bench tooling only; not part of the product path.
"""
import argparse
import hashlib
import sys

import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix(x):
    """splitmix64 finaliser over a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
        return x ^ (x >> np.uint64(31))


def stream(seed, tag, n):
    """n uint64 values of stream `tag` of `seed`."""
    with np.errstate(over="ignore"):
        base = splitmix(np.array([seed * 1000003 + tag], dtype=np.uint64))[0]
        return splitmix(base + np.arange(n, dtype=np.uint64))


ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    COMP[a] = b


def contig_sizes(n, min_len, span):
    return (min_len + (np.arange(n, dtype=np.int64) * 7) % span).astype(np.int64)


def make_reference(n, seed, min_len=104, span=57):
    sizes = contig_sizes(n, min_len, span)
    total = int(sizes.sum())
    seq = ACGT[(stream(seed, 1, total) >> np.uint64(62)).astype(np.int64)]
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    return sizes, starts, seq


def fasta_bytes(sizes, starts, seq):
    parts = []
    for i in range(len(sizes)):
        parts.append(b">t%d\n" % (i + 1))
        parts.append(seq[starts[i]:starts[i + 1]].tobytes())
        parts.append(b"\n")
    return b"".join(parts)


def make_reads(sizes, starts, seq, n_reads, seed, read_len, conv_from, conv_to, p_conv=0.9, max_sub=2):
    """reads spread evenly over the whole contig range (so the last ones lie beyond contig 131 071 when there are more)"""
    r = stream(seed, 2, n_reads * 4).reshape(n_reads, 4)
    ncont = len(sizes)
    out = []
    for i in range(n_reads):
        c = int((i * (ncont - 1)) // max(1, n_reads - 1)) if i % 3 else int(r[i, 0] % np.uint64(ncont))
        room = int(sizes[c]) - read_len
        p = int(r[i, 1] % np.uint64(room + 1))
        s = seq[starts[c] + p: starts[c] + p + read_len].copy()
        rev = bool(r[i, 2] & np.uint64(1))
        if rev:
            s = COMP[s][::-1].copy()
        u = stream(seed, 1000 + i, read_len * 2).reshape(2, read_len)
        conv = (s == ord(conv_from)) & ((u[0] % np.uint64(1000)) < np.uint64(int(p_conv * 1000)))
        s[conv] = ord(conv_to)
        nsub = int(r[i, 3] % np.uint64(max_sub + 1))
        for k in range(nsub):
            q = int(u[1, k] % np.uint64(read_len))
            s[q] = ACGT[(int(np.where(ACGT == s[q])[0][0]) + 1 + int(u[1, k + 8] % np.uint64(3))) % 4] if s[q] in ACGT else s[q]
        out.append((b"r%d_t%d_%d_%s" % (i, c + 1, p + 1, b"-" if rev else b"+"), s.tobytes()))
    return out


def fastq_bytes(reads):
    return b"".join(b"@" + n + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n" for n, s in reads)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contigs", type=int, required=True)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("-M", dest="rule", default="A:G")
    ap.add_argument("--ref-out")
    ap.add_argument("--reads-out")
    a = ap.parse_args()
    sizes, starts, seq = make_reference(a.contigs, a.seed)
    fa = fasta_bytes(sizes, starts, seq)
    if a.ref_out:
        open(a.ref_out, "wb").write(fa)
    if a.reads_out:
        to = a.rule[2] if a.rule[2] in "ACGT" else a.rule[0]
        open(a.reads_out, "wb").write(fastq_bytes(make_reads(sizes, starts, seq, a.reads, a.seed, a.len, a.rule[0], to)))
    sys.stdout.write(hashlib.sha256(fa).hexdigest() + "\n")


if __name__ == "__main__":
    main()
