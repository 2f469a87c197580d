#!/usr/bin/env python3
"""FASTQ -> unaligned, header-less SAM text (the form of SAM the reference can take as read input: a file that starts with an
@HD header reads as FASTQ to its format sniffer, main.cpp:386-405).  With two FASTQ files the mates alternate, flagged 77 / 141,
as reads.cpp:84-110 expects for `-a x.sam -b x.sam`.  Test tooling only."""
import gzip
import sys


def records(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        while True:
            h = f.readline()
            if not h:
                return
            s = f.readline().rstrip("\n")
            f.readline()
            q = f.readline().rstrip("\n")
            yield h[1:].split()[0], s, q


def main():
    fq, out = sys.argv[1], sys.argv[2]
    fq2 = sys.argv[3] if len(sys.argv) > 3 else None
    with open(out, "w") as o:
        if fq2:
            for (n1, s1, q1), (n2, s2, q2) in zip(records(fq), records(fq2)):
                o.write("%s\t77\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (n1, s1, q1))
                o.write("%s\t141\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (n2, s2, q2))
        else:
            for n, s, q in records(fq):
                o.write("%s\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (n, s, q))


if __name__ == "__main__":
    main()
