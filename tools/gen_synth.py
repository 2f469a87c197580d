#!/usr/bin/env python3
"""Seeded synthetic inputs for the BASAL hot-path configs.

Writes a reference FASTA and single-end FASTQ (or a pair of FASTQs) whose reads are
sampled from that reference, base-converted the way the chemistry named by ``-M from:to``
would do it, and mutated.  Everything is driven by ``numpy.random.default_rng(seed)`` so a
fixture can be regenerated bit-for-bit from the command line recorded next to it.

This is test/bench tooling only; it is not part of the product path.
"""
import argparse
import sys

import numpy as np

COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    COMP[a] = b
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_reference(rng, total_bp, n_contigs, n_run_every=0, n_run_len=0, repeat_copies=0,
                   repeat_len=0, lower_frac=0.0):
    """Uniform ACGT contigs; optional N runs, planted repeat family, lower-case stretches."""
    sizes = [total_bp // n_contigs] * n_contigs
    sizes[-1] += total_bp - sum(sizes)
    contigs = []
    family = ACGT[rng.integers(0, 4, size=repeat_len)] if repeat_copies else None
    for ci, sz in enumerate(sizes):
        seq = ACGT[rng.integers(0, 4, size=sz)].copy()
        if repeat_copies:
            per = max(1, repeat_copies // n_contigs)
            for _ in range(per):
                p = int(rng.integers(0, max(1, sz - repeat_len)))
                unit = family.copy()
                # 1 % divergence between copies so the family is a near-repeat, not exact
                mut = rng.random(repeat_len) < 0.01
                unit[mut] = ACGT[rng.integers(0, 4, size=int(mut.sum()))]
                seq[p:p + repeat_len] = unit[: max(0, min(repeat_len, sz - p))]
        if n_run_every:
            for p in range(n_run_every, sz - n_run_len, n_run_every):
                seq[p:p + n_run_len] = ord("N")
        if lower_frac > 0:
            nlow = int(sz * lower_frac / 200)
            for _ in range(nlow):
                p = int(rng.integers(0, max(1, sz - 200)))
                seg = seq[p:p + 200]
                seq[p:p + 200] = np.where(seg != ord("N"), seg | 0x20, seg)
        contigs.append(("chr%d" % (ci + 1), seq))
    return contigs


def write_fasta(path, contigs, width=60):
    with open(path, "wb") as f:
        for name, seq in contigs:
            f.write(b">" + name.encode() + b"\n")
            b = seq.tobytes()
            for i in range(0, len(b), width):
                f.write(b[i:i + width] + b"\n")


def convert(rng, frag, frm, tos, p_conv):
    """Apply the chemistry on the sequenced strand: each `frm` base becomes one of `tos`
    with probability p_conv; '-' among `tos` deletes the base."""
    frag = frag.copy()
    is_from = (frag == ord(frm))
    hit = is_from & (rng.random(frag.size) < p_conv)
    if not hit.any():
        return frag
    choice = rng.integers(0, len(tos), size=frag.size)
    keep = np.ones(frag.size, dtype=bool)
    for k, t in enumerate(tos):
        sel = hit & (choice == k)
        if t == "-":
            keep &= ~sel
        else:
            frag[sel] = ord(t)
    return frag[keep]


def mutate(rng, read, n_sub=None, sub_rate=0.0):
    read = read.copy()
    if n_sub is None:
        pos = np.nonzero(rng.random(read.size) < sub_rate)[0]
    else:
        pos = rng.choice(read.size, size=min(n_sub, read.size), replace=False) if n_sub else []
    for p in pos:
        old = read[p]
        alts = [c for c in b"ACGT" if c != (old & 0xDF)]
        read[p] = alts[int(rng.integers(0, 3))]
    return read


def add_indel(rng, read, max_len):
    k = int(rng.integers(1, max_len + 1))
    p = int(rng.integers(10, max(11, read.size - 10 - k)))
    if rng.random() < 0.5:  # insertion on the read
        ins = ACGT[rng.integers(0, 4, size=k)]
        return np.concatenate([read[:p], ins, read[p:]])
    return np.concatenate([read[:p], read[p + k:]])  # deletion from the read


def sample_fragment(rng, contigs, length, upper=True):
    for _ in range(1000):
        ci = int(rng.integers(0, len(contigs)))
        seq = contigs[ci][1]
        if seq.size <= length:
            continue
        p = int(rng.integers(0, seq.size - length))
        frag = seq[p:p + length]
        if (frag == ord("N")).sum() > 2:
            continue
        return ci, p, (frag & 0xDF if upper else frag)
    raise RuntimeError("could not sample a fragment")


def revcomp(a):
    return COMP[a[::-1]]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ref-out")
    ap.add_argument("--ref-in", help="use an existing FASTA instead of generating one")
    ap.add_argument("--ref-bp", type=int, default=1_000_000)
    ap.add_argument("--contigs", type=int, default=2)
    ap.add_argument("--ref-seed", type=int, default=1)
    ap.add_argument("--n-run-every", type=int, default=0)
    ap.add_argument("--n-run-len", type=int, default=0)
    ap.add_argument("--repeat-copies", type=int, default=0)
    ap.add_argument("--repeat-len", type=int, default=0)
    ap.add_argument("--lower-frac", type=float, default=0.0)
    ap.add_argument("--reads-out", help="FASTQ for SE / mate 1")
    ap.add_argument("--reads2-out", help="FASTQ for mate 2 (turns on PE)")
    ap.add_argument("--reads", type=int, default=1000)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--len-jitter", type=int, default=0, help="read length drawn from [len-jitter, len]")
    ap.add_argument("--reads-seed", type=int, default=2)
    ap.add_argument("-M", dest="rule", default="C:T")
    ap.add_argument("--p-conv", type=float, default=0.95)
    ap.add_argument("--max-sub", type=int, default=3, help="0..max-sub substitutions/read (uniform)")
    ap.add_argument("--sub-rate", type=float, default=None, help="per-base substitution rate instead of --max-sub")
    ap.add_argument("--indel-frac", type=float, default=0.0)
    ap.add_argument("--indel-max", type=int, default=2)
    ap.add_argument("--rev-frac", type=float, default=0.5)
    ap.add_argument("--n-frac", type=float, default=0.0, help="fraction of reads given 1-7 N's")
    ap.add_argument("--junk-frac", type=float, default=0.0, help="fraction of reads that are random sequence")
    ap.add_argument("--frag-min", type=int, default=200)
    ap.add_argument("--frag-max", type=int, default=600)
    ap.add_argument("--fasta-reads", action="store_true", help="write reads as FASTA instead of FASTQ")
    ap.add_argument("--lower-reads-frac", type=float, default=0.0, help="fraction of reads written in lower case")
    ap.add_argument("--iupac-frac", type=float, default=0.0, help="fraction of reads given 1-3 IUPAC ambiguity codes (R, Y, K ...)")
    ap.add_argument("--pbat", action="store_true", help="emit the reverse complement of each SE read (PBAT, -n 2)")
    args = ap.parse_args(argv)

    frm, tos = args.rule.split(":")
    frm = frm.upper()
    tos = [t for t in tos.upper()]

    if args.ref_in:
        contigs = []
        name, chunks = None, []
        with open(args.ref_in, "rb") as f:
            for line in f:
                if line.startswith(b">"):
                    if name is not None:
                        contigs.append((name, np.frombuffer(b"".join(chunks), dtype=np.uint8)))
                    name, chunks = line[1:].split()[0].decode(), []
                else:
                    chunks.append(line.strip())
        contigs.append((name, np.frombuffer(b"".join(chunks), dtype=np.uint8)))
    else:
        rng = np.random.default_rng(args.ref_seed)
        contigs = make_reference(rng, args.ref_bp, args.contigs, args.n_run_every, args.n_run_len,
                                 args.repeat_copies, args.repeat_len, args.lower_frac)
        if args.ref_out:
            write_fasta(args.ref_out, contigs)
    if not args.reads_out:
        return 0

    rng = np.random.default_rng(args.reads_seed)
    pe = bool(args.reads2_out)
    f1 = open(args.reads_out, "wb")
    f2 = open(args.reads2_out, "wb") if pe else None

    def emit(f, name, seq):
        if args.fasta_reads:
            f.write(b">" + name + b"\n" + seq.tobytes() + b"\n")
        else:
            f.write(b"@" + name + b"\n" + seq.tobytes() + b"\n+\n" + b"I" * seq.size + b"\n")

    for r in range(args.reads):
        rl = args.len - (int(rng.integers(0, args.len_jitter + 1)) if args.len_jitter else 0)
        name = b"r%d" % r
        if rng.random() < args.junk_frac:
            emit(f1, name + b"_junk", ACGT[rng.integers(0, 4, size=rl)])
            if pe:
                emit(f2, name + b"_junk", ACGT[rng.integers(0, 4, size=rl)])
            continue
        if pe:
            fl = int(rng.integers(args.frag_min, args.frag_max + 1))
            ci, p, frag = sample_fragment(rng, contigs, fl + 8)
        else:
            ci, p, frag = sample_fragment(rng, contigs, rl + 8)
        rev = rng.random() < args.rev_frac
        if rev:
            frag = revcomp(frag)
        # chemistry acts on the sequenced strand: revcomp THEN convert (directional protocol)
        frag = convert(rng, frag, frm, tos, args.p_conv)
        tag = name + b"_%s_%d_%s" % (contigs[ci][0].encode(), p + 1, b"-" if rev else b"+")
        m1 = frag[:rl]
        if args.sub_rate is not None:
            m1 = mutate(rng, m1, None, args.sub_rate)
        else:
            m1 = mutate(rng, m1, int(rng.integers(0, args.max_sub + 1)))
        if rng.random() < args.indel_frac:
            m1 = add_indel(rng, m1, args.indel_max)[:rl]
        if rng.random() < args.n_frac:
            k = int(rng.integers(1, 8))
            m1 = m1.copy()
            m1[rng.choice(m1.size, size=k, replace=False)] = ord("N")
        if args.iupac_frac > 0 and rng.random() < args.iupac_frac:  # (no draw when the option is off: older fixtures stay reproducible)
            m1 = m1.copy()
            for q in rng.choice(m1.size, size=int(rng.integers(1, 4)), replace=False):
                m1[q] = b"RYKMSW"[int(rng.integers(0, 6))]
        if args.lower_reads_frac > 0 and rng.random() < args.lower_reads_frac:
            m1 = m1 | 0x20
        emit(f1, tag, revcomp(m1) if args.pbat else m1)
        if pe:
            m2 = revcomp(frag)[:rl]
            if args.sub_rate is not None:
                m2 = mutate(rng, m2, None, args.sub_rate)
            else:
                m2 = mutate(rng, m2, int(rng.integers(0, args.max_sub + 1)))
            emit(f2, tag, m2)
    f1.close()
    if f2:
        f2.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
