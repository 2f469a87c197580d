"""FASTQ -> unaligned BAM (BGZF), for the BAM-input fixtures.  usage: fq2bam.py in.fq[.gz] out.bam [in2.fq[.gz]]
With a second FASTQ the mates are interleaved (flags 0x4D / 0x8D), the layout the reference's PE BAM reader expects
(reads.cpp:84-110: mate 1 = every first record, mate 2 = every second).  Plain zlib + struct: no samtools needed."""
import gzip
import struct
import sys
import zlib

NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def fastq(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        while True:
            h = f.readline()
            if not h:
                return
            s = f.readline().strip()
            f.readline()
            q = f.readline().strip()
            yield h[1:].split()[0], s, q


def record(name, seq, qual, flag):
    n = len(seq)
    packed = bytearray((n + 1) // 2)
    for i, c in enumerate(seq.upper()):
        packed[i >> 1] |= NT16.get(c, 15) << (4 if i % 2 == 0 else 0)
    q = bytes(ord(c) - 33 for c in qual)
    nm = name.encode() + b"\0"
    body = struct.pack("<iiBBHHHiiii", -1, -1, len(nm), 0, 4680, 0, flag, n, -1, -1, 0) + nm + bytes(packed) + q
    return struct.pack("<i", len(body)) + body


def bgzf_block(data):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def main():
    a = list(fastq(sys.argv[1]))
    b = list(fastq(sys.argv[3])) if len(sys.argv) > 3 else None
    text = b"@HD\tVN:1.0\tSO:unsorted\n"
    raw = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 0))
    for i, (n, s, q) in enumerate(a):
        raw += record(n, s, q, 0x4D if b else 0x4)
        if b:
            n2, s2, q2 = b[i]
            raw += record(n2, s2, q2, 0x8D)
    with open(sys.argv[2], "wb") as o:
        for p in range(0, len(raw), 60000):
            o.write(bgzf_block(bytes(raw[p:p + 60000])))
        o.write(bgzf_block(b""))


if __name__ == "__main__":
    main()
