"""Genome-scale FASTA / FASTQ FILES for the command-line tests and benchmarks: tools/synth_gpu.py generates the genome and the
reads on the GPU, this module adds what real read files have (variable lengths, N's, low-quality tails, adapter read-through,
lower-case bases) and writes the text.  Everything is seeded.  Test/bench tooling only; not part of the product path."""
import numpy as np

ADAPTER = b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"


def write_fasta(path, G, width=60):
    """G: synth_gpu.Genome (base ids per contig on the GPU, N runs)."""
    asc = np.frombuffer(b"ACGT", np.uint8)
    if getattr(G, "flat", False):  # synth_gpu.make_transcriptome: one flat base array, no N's
        flat = asc[G.ids.cpu().numpy()]
        off = G.base_off.cpu().numpy()
        with open(path, "wb") as f:
            for name, o, n in zip(G.names, off, G.sizes):
                s = flat[o:o + n]
                full = n // width
                body = np.empty((full, width + 1), np.uint8)
                body[:, :width] = s[: full * width].reshape(full, width)
                body[:, width] = ord("\n")
                f.write(b">" + name.encode() + b"\n" + body.tobytes() + (s[full * width:].tobytes() + b"\n" if n % width else b""))
        return
    with open(path, "wb") as f:
        for name, ids, runs in zip(G.names, G.ids, G.nmask_runs):
            s = asc[ids.cpu().numpy()]
            for a, b in runs:
                s[a:b] = ord("N")
            f.write(b">" + name.encode() + b"\n")
            n = len(s)
            full = n // width
            body = np.empty((full, width + 1), np.uint8)
            body[:, :width] = s[: full * width].reshape(full, width)
            body[:, width] = ord("\n")
            f.write(body.tobytes())
            if n % width:
                f.write(s[full * width:].tobytes() + b"\n")


def fastq_bytes(seqs, lens, quals=None, first=0, name_prefix=b"r", fasta=False):
    """seqs: (n, L) uint8 ASCII; lens: (n,) read lengths <= L; quals: (n, L) uint8 or None ('I')."""
    n, L = seqs.shape
    lens = np.asarray(lens, np.int64)
    ids = np.arange(first, first + n)
    width = len(str(first + n))
    name = np.frombuffer(b"".join(b"%0*d" % (width, i) for i in ids), np.uint8).reshape(n, width)
    col = np.arange(L)[None, :]
    keep = col < lens[:, None]
    parts = [np.full((n, 1), ord(">" if fasta else "@"), np.uint8), np.tile(np.frombuffer(name_prefix, np.uint8), (n, 1)), name,
             np.full((n, 1), 10, np.uint8), np.where(keep, seqs, 0).astype(np.uint8), np.full((n, 1), 10, np.uint8)]
    if not fasta:
        q = quals if quals is not None else np.full((n, L), ord("I"), np.uint8)
        parts += [np.tile(np.frombuffer(b"+\n", np.uint8), (n, 1)), np.where(keep, q, 0).astype(np.uint8), np.full((n, 1), 10, np.uint8)]
    flat = np.concatenate(parts, axis=1).reshape(-1)
    return flat[flat != 0].tobytes()


def dirty(seqs, seed, min_len=None, n_rate=0.002, many_n_frac=0.01, lowq_frac=0.2, adapter_frac=0.1, lower_frac=0.05):
    """Returns (seqs', lens, quals): reads as a sequencer would deliver them."""
    rng = np.random.default_rng(seed)
    n, L = seqs.shape
    s = seqs.copy()
    lens = np.full(n, L, np.int64) if min_len is None else rng.integers(min_len, L + 1, size=n)
    q = np.full((n, L), ord("I"), np.uint8)
    s[rng.random((n, L)) < n_rate] = ord("N")
    many = rng.random(n) < many_n_frac
    s[many[:, None] & (rng.random((n, L)) < 0.15)] = ord("N")
    # adapter read-through: the insert ends at `ins`, the adapter follows
    ad = np.frombuffer(ADAPTER, np.uint8)
    has_ad = rng.random(n) < adapter_frac
    ins = rng.integers(30, L, size=n)
    col = np.arange(L)[None, :]
    rel = col - ins[:, None]
    in_ad = has_ad[:, None] & (rel >= 0) & (rel < len(ad))
    s[in_ad] = ad[np.clip(rel, 0, len(ad) - 1)][in_ad]
    past = has_ad[:, None] & (rel >= len(ad))
    s[past] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=int(past.sum()))]
    # low-quality tails
    lowq = rng.random(n) < lowq_frac
    tail = rng.integers(1, 40, size=n)
    q[lowq[:, None] & (col >= (lens - tail)[:, None])] = ord("#")
    lower = rng.random(n) < lower_frac
    s[lower] = s[lower] | 0x20
    return s, lens, q
