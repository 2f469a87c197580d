"""The batch pipeline (basal_pipe_*): raw read text in, SAM text out, everything in between on the GPU (text parsing, FilterReads,
inherited-state table, alignment, SAM assembly).  Checked against the golden SAMs the reference binary printed."""
import gzip
import os

import numpy as np
import pytest

import basal_amd as B
import harness as H

pytestmark = pytest.mark.gpu


def staged_core(name, extra_flags=()):
    fa, fq, _, _ = H.fixture_paths(name)
    flags = H.MANIFEST[name]["flags"] + list(extra_flags)
    p = B.Params(H.rule_of(flags), flags)
    ref = B.Reference(p, fasta_path=fa)
    core = B.Core(p)
    core.upload(ref, build_on_gpu=True)
    core.set_contig_names(ref.names())
    return p, ref, core, fq


def read_text(path):
    return gzip.open(path, "rb").read() if path.endswith(".gz") else open(path, "rb").read()


def split_records(text, lines_per_record, records_per_batch):
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    step = lines_per_record * records_per_batch
    return [b"\n".join(lines[i:i + step]) + b"\n" for i in range(0, len(lines), step)]


def run_text(name, records_per_batch=None, depth=3):
    p, ref, core, fq = staged_core(name)
    text = read_text(fq)
    fasta = text[:1] == b">"
    batches = split_records(text, 2 if fasta else 4, records_per_batch) if records_per_batch else [text]
    pipe = B.Pipe(core, depth=depth, max_reads=8192, max_bytes=4 << 20)
    out, stats = [], []
    pending = 0
    for b in batches:
        if pending == depth:  # a single thread drives both ends: make room first
            rc, data, st = pipe.collect()
            assert rc == 0, data
            out.append(data); stats.append(st); pending -= 1
        pipe.submit_text(b, B.FMT_FASTA if fasta else B.FMT_FASTQ)
        pending += 1
    while pending:
        rc, data, st = pipe.collect()
        assert rc == 0, data
        out.append(data); stats.append(st); pending -= 1
    pipe.close()
    return H.sam_header(ref) + b"".join(out).decode(), stats


@pytest.mark.parametrize("name", H.SE)
def test_pipe_text_sam_matches_golden(name):
    got, stats = run_text(name)
    assert got == H.golden_sam(name)
    import oracle as orc
    assert sum(s.n_reads for s in stats) == len(orc.read_fastx(H.fixture_paths(name)[1]))


@pytest.mark.parametrize("name", ["varlen_trim", "varlen_s16", "rep_r2_w10", "ct_g3", "I3", "edge_short", "len_bound_128", "long_300", "fa_reads"])
@pytest.mark.parametrize("per_batch", [1, 7, 64])
def test_pipe_small_batches(name, per_batch):
    """Batch boundaries must not show: the state later reads inherit from earlier ones lives on the device across batches."""
    got, _ = run_text(name, records_per_batch=per_batch, depth=2 if per_batch == 1 else 3)
    assert got == H.golden_sam(name)


def make_records(reads, first_index=0, readset=0):
    """(blob, raw table) the way a host-side decoder would hand reads over: name, bases, qualities back to back, no separators."""
    blob = bytearray()
    raw = np.zeros(len(reads), B.RAWREAD_DTYPE)
    for i, (n, s, q) in enumerate(reads):
        r = raw[i]
        r["name_off"], r["name_len"] = len(blob), len(n)
        blob += n.encode()
        r["seq_off"], r["seq_len"] = len(blob), len(s)
        blob += s.encode()
        r["qual_off"], r["qual_len"] = len(blob), len(q)
        blob += q.encode()
        r["readset"], r["index"] = readset, first_index + i
    return bytes(blob), raw


@pytest.mark.parametrize("name", ["ct_basic", "ct_n1_dirty", "varlen_trim", "tdel_pipeline", "rep_r2_w10", "contigs_140k_g1", "fa_reads"])
def test_pipe_records_sam_matches_golden(name):
    import oracle as orc
    p, ref, core, fq = staged_core(name)
    reads = orc.read_fastx(fq)
    fasta = read_text(fq)[:1] == b">"
    reads = [(n, s[: p.c.max_readlen], "" if fasta else (q[: p.c.max_readlen] if len(s) > p.c.max_readlen else q)) for n, s, q in reads]
    pipe = B.Pipe(core, depth=2, max_reads=8192, max_bytes=4 << 20)
    out = []
    for b0 in range(0, len(reads), 200):
        blob, raw = make_records(reads[b0:b0 + 200], first_index=b0)
        pipe.submit_records(blob, raw)
        rc, data, st = pipe.collect()
        assert rc == 0, data
        out.append(data)
    assert H.sam_header(ref) + b"".join(out).decode() == H.golden_sam(name)


def test_pipe_irregular_text_is_refused_and_resubmitted():
    """Text that line parsing and token parsing read differently is handed back (BASAL_EIO) with the later batches dropped."""
    import oracle as orc
    name = "ct_basic"
    p, ref, core, fq = staged_core(name)
    text = read_text(fq)
    batches = split_records(text, 4, 100)
    pipe = B.Pipe(core, depth=3, max_reads=8192, max_bytes=4 << 20)
    bad = batches[1].replace(b"\n+\n", b"\n\n+\n", 1)  # a blank line: the reference's reader skips it, a line parser must not guess
    pipe.submit_text(batches[0])
    pipe.submit_text(bad)
    pipe.submit_text(batches[2])
    rc, d0, _ = pipe.collect()
    assert rc == 0
    rc, msg, _ = pipe.collect()
    assert rc == -6 and "regular" in msg  # BASAL_EIO
    rc, msg, _ = pipe.collect()
    assert rc == -4 and "stopped" in msg  # BASAL_ESTATE until the caller rewinds; the batch behind the refused one is dropped then
    pipe.rewind()
    # the caller parses the refused batch itself and goes on from there
    reads = orc.read_fastx(fq)
    blob, raw = make_records(reads[100:200], first_index=100)
    pipe.submit_records(blob, raw)
    out = [d0]
    rc, d, _ = pipe.collect()
    assert rc == 0
    out.append(d)
    for b in batches[2:]:
        pipe.submit_text(b)
        rc, d, _ = pipe.collect()
        assert rc == 0
        out.append(d)
    assert H.sam_header(ref) + b"".join(out).decode() == H.golden_sam(name)


def test_pipe_prepared_results_equal_align_batch():
    """BASAL_PIPE_OUT_RESULTS with prepared reads = basal_core_align_batch, through pinned buffers and the slot streams."""
    import oracle as orc
    name = "c1_s16"
    fa, fq, _, _ = H.fixture_paths(name)
    flags = H.MANIFEST[name]["flags"]
    p = B.Params(H.rule_of(flags), flags)
    ref = B.Reference(p, fasta_path=fa)
    core = B.Core(p)
    core.upload(ref, build_on_gpu=True)
    recs = H.filter_reads(p, orc.read_fastx(fq))
    bases, descs, stales = H.make_batch(p, recs)
    assert len(stales) == 0
    want, _, _ = core.align_batch(bases, descs)
    pipe = B.Pipe(core, depth=2, max_reads=4096, max_bytes=1 << 20, output=B.PIPE_OUT_RESULTS)
    got = []
    for b0 in range(0, len(descs), 300):
        d = descs[b0:b0 + 300].copy()
        live = d["len"] > 0
        lo, hi = int(d["seq_off"][live].min()), int((d["seq_off"][live] + d["len"][live]).max())
        d["seq_off"][live] -= lo
        pipe.submit_prepared(bases[lo:hi], d, 128)
        rc, data, st = pipe.collect()
        assert rc == 0, data
        got.append(np.frombuffer(data, B.core.RESULT_DTYPE))
    got = np.concatenate(got)
    assert got.tobytes() == want.tobytes()
