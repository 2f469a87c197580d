"""Shared test plumbing: fixture manifest, and a driver that pushes a fixture through the PRODUCT
(libbasal_amd.so via ctypes: host filter -> GPU core -> host SAM formatter)."""
import ctypes as C
import gzip
import json
import os

import numpy as np

import basal_amd as B
from basal_amd import core as bc
import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))
SE = sorted(k for k, v in MANIFEST.items() if not v["pe"])
PE = sorted(k for k, v in MANIFEST.items() if v["pe"])


def _generated_fasta(name, g):
    """A reference too large to commit (tools/gen_contigs.py): regenerated once per checkout into tests/golden/_gen/ and checked
    against the SHA-256 recorded when the reference binary printed the golden SAM for it."""
    import hashlib
    import sys
    path = os.path.join(GOLD, "_gen", name + ".fa")
    if not os.path.exists(path):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import gen_contigs as gc
        fa = gc.fasta_bytes(*gc.make_reference(g["contigs"], g["seed"]))
        assert hashlib.sha256(fa).hexdigest() == g["sha256"], "regenerated FASTA of %s differs from the one the golden SAM was made on" % name
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path + ".tmp%d" % os.getpid(), "wb") as f:
            f.write(fa)
        os.replace(path + ".tmp%d" % os.getpid(), path)
    return path


def fixture_paths(name):
    m = MANIFEST[name]
    fa = _generated_fasta(name, m["fasta_gen"]) if m.get("fasta_gen") else os.path.join(GOLD, name + ".fa.gz")
    return (fa, os.path.join(GOLD, m["reads_file"]),
            os.path.join(GOLD, name + "_2.fq.gz") if m["pe"] else None, os.path.join(GOLD, name + ".sam.gz"))


def golden_sam(name):
    return gzip.open(fixture_paths(name)[3], "rb").read().decode()


def rule_of(flags):
    return flags[flags.index("-M") + 1]


def filter_reads(params, reads, readset=0, first_index=0):
    """Product-side FilterReads for a list of (name, seq, qual). Returns list of dicts."""
    L = B.lib()
    out = []
    for i, (name, seq, qual) in enumerate(reads):
        if len(seq) > params.c.max_readlen:
            seq, qual = seq[: params.c.max_readlen], qual[: params.c.max_readlen]
        sb = C.create_string_buffer(seq.encode(), len(seq) + 2)
        qb = C.create_string_buffer(qual.encode(), max(len(seq), len(qual)) + 2)
        ms = C.c_uint32()
        qc = L.basal_host_filter_read(C.byref(params.c), sb, qb, C.byref(ms))
        out.append({"name": name, "seq": sb.value.decode(), "qual": qb.value.decode(), "qc": qc, "max_snp": ms.value,
                    "index": first_index + i, "readset": readset})
    return out


class StaleTracker:
    """basal_host_stale_* (the host-side tracker of state inherited between reads)."""

    def __init__(self, params):
        self.h = B.lib().basal_host_stale_new(C.byref(params.c))

    def __del__(self):
        try:
            B.lib().basal_host_stale_free(self.h)
        except Exception:
            pass


def make_batch(params, recs, tracker=None):
    """(bases, descriptors, stale table) for basal_core_align_batch."""
    L = B.lib()
    tracker = tracker or StaleTracker(params)
    L.basal_host_stale_begin_batch(tracker.h)
    descs = np.zeros(len(recs), bc.READ_DTYPE)
    chunks, stales = [], []
    off = 0
    for i, r in enumerate(recs):
        d = descs[i]
        d["index"], d["readset"], d["stale_idx"] = r["index"], r["readset"], B.STALE_NONE
        if r["qc"]:
            continue
        n = len(r["seq"])
        d["len"], d["max_snp"], d["seq_off"] = n, r["max_snp"], off
        chunks.append(np.frombuffer(r["seq"].encode(), np.uint8))
        off += n
        se = bc.basal_stale()
        if L.basal_host_stale_visit(tracker.h, r["seq"].encode(), n, r["readset"], 0, i, C.byref(se)):
            d["stale_idx"] = len(stales)
            stales.append(np.frombuffer(bytes(se), bc.STALE_DTYPE)[0])
    bases = np.concatenate(chunks) if chunks else np.zeros(1, np.uint8)
    st = np.array(stales, dtype=bc.STALE_DTYPE) if stales else np.zeros(0, bc.STALE_DTYPE)
    return bases, descs, st


def format_se(params, ref, recs, results, stream):
    L = B.lib()
    out = []
    buf = C.create_string_buffer(1 << 20)
    sp = stream.ctypes.data if len(stream) else None
    for r, res in zip(recs, results):
        rs = bc.basal_result.from_buffer_copy(res.tobytes())
        n = L.basal_host_format_se(C.byref(params.c), ref.h, r["name"].encode(), r["seq"].encode(), r["qual"].encode(),
                                   r["readset"], int(r["qc"]), C.byref(rs), sp, buf, len(buf))
        assert n >= 0, L.basal_last_error()
        out.append(buf.raw[:n].decode())
    return out


def sam_header(ref):
    buf = C.create_string_buffer((1 << 20) + 160 * ref.ncontig)
    n = B.lib().basal_host_sam_header(ref.h, b"x", buf, len(buf))
    assert n >= 0, B.lib().basal_last_error()
    return "".join(l + "\n" for l in buf.raw[:n].decode().splitlines() if not l.startswith("@PG"))


def hit_tuple(h):
    return (int(h["loc"]), int(h["chr"]), int(h["gap_size"]), int(h["strand"]), int(h["gap_pos"]), int(h["level"]),
            int(h["chain"]), int(h["mode"]))
