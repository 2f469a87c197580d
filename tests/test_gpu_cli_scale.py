"""The `basal` command line at scale, end to end, against the REFERENCE binary run with -p 1 on the same files (oracle/_ref/basal
travels with the snapshot; the CPU oracle's CLI stands in if it is missing): SAM text identical byte for byte (minus @PG) on
read files with what real files have -- N's, low-quality tails, adapter read-through, lower case, variable lengths, repeats."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import basal_amd as B
import harness as H
import oracle as orc

sys.path.insert(0, os.path.join(H.ROOT, "tools"))
pytestmark = pytest.mark.gpu
BASAL_BIN = os.path.join(H.ROOT, "basal_amd", "bin", "basal")


def sam_digest(path):
    h = hashlib.md5()
    n = 0
    with open(path, "rb") as f:
        for line in f:
            if not line.startswith(b"@PG"):
                h.update(line)
                n += not line.startswith(b"@")
    return h.hexdigest(), n


def make_files(tmp, rule, n_reads, read_len, seed, scale=0.02, repeat_copies=0, min_len=None, p_conv=0.95, **dirt):
    import torch
    import synth_gpu
    import synth_files
    dev = torch.device("cuda", 0)
    p = B.Params(rule, ["-M", rule])
    G = synth_gpu.make_genome(p, dev, scale=scale, seed=seed, repeat_copies=repeat_copies)
    fa, fq = os.path.join(tmp, "g.fa"), os.path.join(tmp, "r.fq")
    synth_files.write_fasta(fa, G)
    frm = "ACGT".index(rule[0])
    tos = [t for t in rule[2:] if t in "ACGT"]
    to = "ACGT".index(tos[0]) if tos else frm
    bases, _, _, _ = synth_gpu.make_reads(G, n_reads, dev, read_len=read_len, seed=seed + 1, conv_from=frm, conv_to=to, p_conv=p_conv if tos else 0.0)
    seqs = bases.cpu().numpy().reshape(n_reads, read_len)
    s, lens, q = synth_files.dirty(seqs, seed + 2, min_len=min_len, **dirt)
    with open(fq, "wb") as f:
        for b0 in range(0, n_reads, 250_000):
            f.write(synth_files.fastq_bytes(s[b0:b0 + 250_000], lens[b0:b0 + 250_000], q[b0:b0 + 250_000], first=b0))
    del G
    torch.cuda.empty_cache()
    return fa, fq


def run_both(tmp, fa, fq, flags, env=None):
    out, ref = os.path.join(tmp, "out.sam"), os.path.join(tmp, "ref.sam")
    r = subprocess.run([BASAL_BIN, "-a", fq, "-d", fa] + flags + ["-p", "8", "-o", out], capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr
    checker = orc.REF_BIN if os.path.exists(orc.REF_BIN) else orc.CLI
    c = subprocess.run([checker, "-a", fq, "-d", fa] + flags + ["-p", "1", "-o", ref], capture_output=True, text=True)
    assert c.returncode == 0, c.stderr
    got, want = sam_digest(out), sam_digest(ref)
    assert got[1] == want[1], "record counts differ: %d vs %d" % (got[1], want[1])
    assert got[0] == want[0], "SAM text differs from %s" % os.path.basename(checker)
    return r.stderr, got[1]


def test_cli_1m_dirty_reads_match_reference(tmp_path):
    fa, fq = make_files(str(tmp_path), "C:T", 1_000_000, 100, seed=21)
    log, n = run_both(str(tmp_path), fa, fq, ["-M", "C:T", "-S", "1", "-s", "12", "-u", "-q", "10", "-A", "AGATCGGAAGAGC"])
    assert n == 1_000_000
    assert "parsing it on the host" not in log


def test_cli_repeats_r2_match_reference(tmp_path):
    fa, fq = make_files(str(tmp_path), "C:T", 200_000, 100, seed=31, repeat_copies=200_000, n_rate=0.0, many_n_frac=0.0, lowq_frac=0.0, adapter_frac=0.0, lower_frac=0.0)
    run_both(str(tmp_path), fa, fq, ["-M", "C:T", "-S", "3", "-s", "12", "-r", "2", "-w", "20", "-n", "1", "-k", "1e-3"])


def test_cli_varlen_multiway_gap_match_reference(tmp_path):
    fa, fq = make_files(str(tmp_path), "A:CGT", 200_000, 150, seed=41, min_len=40, p_conv=0.3)
    run_both(str(tmp_path), fa, fq, ["-M", "A:CGT", "-S", "1", "-s", "12", "-g", "2", "-R", "-u"])


def test_cli_many_small_text_batches(tmp_path):
    """The reader's record cut and the device-side carry across ~1000 batches of a plain text file."""
    fa, fq = make_files(str(tmp_path), "C:T", 60_000, 120, seed=51, min_len=36)
    log, n = run_both(str(tmp_path), fa, fq, ["-M", "C:T", "-S", "1", "-s", "12", "-u"], env={"BASAL_PIPE_BYTES": "16384"})
    assert n == 60_000


@pytest.mark.parametrize("name", ["ct_basic", "varlen_trim", "edge_short", "fa_reads", "rep_r2_w10", "long_490_g1"])
def test_cli_plain_text_input_small_batches(name, tmp_path):
    """Uncompressed read files take the text path (the GPU finds the records); tiny batches exercise the reader's cut."""
    import gzip
    fa, fq, _, _ = H.fixture_paths(name)
    plain = str(tmp_path / "reads.txt")
    open(plain, "wb").write(gzip.open(fq, "rb").read())
    out = str(tmp_path / "o.sam")
    r = subprocess.run([BASAL_BIN, "-a", plain, "-d", fa] + H.MANIFEST[name]["flags"] + ["-o", out], capture_output=True, text=True,
                       env=dict(os.environ, BASAL_PIPE_BYTES="8192"))
    assert r.returncode == 0, r.stderr
    assert "parsing it on the host" not in r.stderr
    got = "".join(l for l in open(out) if not l.startswith("@PG"))
    assert got == H.golden_sam(name)


def test_cli_irregular_text_falls_back_to_host_parsing(tmp_path):
    """Blank lines between records: the reference's token reader takes them in its stride; the text path refuses the batch and the
    CLI goes on parsing on the host from that batch's first byte -- same SAM."""
    import gzip
    name = "ct_basic"
    fa, fq, _, _ = H.fixture_paths(name)
    lines = gzip.open(fq, "rb").read().split(b"\n")
    lines.insert(4 * 137, b"")  # after 137 records
    plain = str(tmp_path / "reads.fq")
    open(plain, "wb").write(b"\n".join(lines))
    out = str(tmp_path / "o.sam")
    r = subprocess.run([BASAL_BIN, "-a", plain, "-d", fa] + H.MANIFEST[name]["flags"] + ["-o", out], capture_output=True, text=True,
                       env=dict(os.environ, BASAL_PIPE_BYTES="8192"))
    assert r.returncode == 0, r.stderr
    assert "parsing it on the host" in r.stderr
    got = "".join(l for l in open(out) if not l.startswith("@PG"))
    assert got == H.golden_sam(name)


def test_cli_pe_200k_pairs_match_reference(tmp_path):
    """BASELINE.json config 3's reads (150-base pairs, -M A:G) at 200 000 pairs on a 24-contig reference: the CLI (mates aligned and
    paired on the GPU) against the reference binary, byte for byte.  (The many-contig reference shapes follow below.)"""
    import torch
    import synth_gpu
    import synth_files
    tmp = str(tmp_path)
    dev = torch.device("cuda", 0)
    p = B.Params("A:G", ["-M", "A:G"])
    G = synth_gpu.make_genome(p, dev, scale=0.01, seed=61, repeat_copies=2000)
    fa, f1, f2 = os.path.join(tmp, "g.fa"), os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")
    synth_files.write_fasta(fa, G)
    n, L = 200_000, 150
    b1, b2 = synth_gpu.make_pairs(G, n, dev, read_len=L, seed=62)
    s1, s2 = b1.cpu().numpy().reshape(n, L), b2.cpu().numpy().reshape(n, L)
    d1, l1, q1 = synth_files.dirty(s1, 63, adapter_frac=0.0, lower_frac=0.0)
    d2, l2, q2 = synth_files.dirty(s2, 64, adapter_frac=0.0, lower_frac=0.0)
    open(f1, "wb").write(synth_files.fastq_bytes(d1, l1, q1, name_prefix=b"p"))
    open(f2, "wb").write(synth_files.fastq_bytes(d2, l2, q2, name_prefix=b"p"))
    del G
    torch.cuda.empty_cache()
    flags = ["-M", "A:G", "-S", "1", "-s", "12", "-u", "-x", "700"]
    out, ref = os.path.join(tmp, "out.sam"), os.path.join(tmp, "ref.sam")
    r = subprocess.run([BASAL_BIN, "-a", f1, "-b", f2, "-d", fa] + flags + ["-p", "16", "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    checker = orc.REF_BIN if os.path.exists(orc.REF_BIN) else orc.CLI
    # (relative paths: the reference's paired-end branch sprintf()s its command line into a 256-byte buffer, main.cpp:410,522)
    c = subprocess.run([checker, "-a", "r1.fq", "-b", "r2.fq", "-d", "g.fa"] + flags + ["-p", "1", "-o", "ref.sam"], capture_output=True, text=True, cwd=tmp)
    assert c.returncode == 0, c.stderr
    got, want = sam_digest(out), sam_digest(ref)
    assert got[1] == want[1] and got[1] >= 2 * n * 0.9
    assert got[0] == want[0]


def _pe_files(tmp, G, n, L, seed):
    import synth_files
    import synth_gpu
    import torch
    fa, f1, f2 = os.path.join(tmp, "g.fa"), os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")
    synth_files.write_fasta(fa, G)
    b1, b2 = synth_gpu.make_pairs(G, n, torch.device("cuda", 0), read_len=L, seed=seed)
    s1, s2 = b1.cpu().numpy().reshape(n, L), b2.cpu().numpy().reshape(n, L)
    d1, l1, q1 = synth_files.dirty(s1, seed + 1, adapter_frac=0.0, lower_frac=0.0)
    d2, l2, q2 = synth_files.dirty(s2, seed + 2, adapter_frac=0.0, lower_frac=0.0)
    open(f1, "wb").write(synth_files.fastq_bytes(d1, l1, q1, name_prefix=b"p"))
    open(f2, "wb").write(synth_files.fastq_bytes(d2, l2, q2, name_prefix=b"p"))
    return fa, f1, f2


def test_cli_pe_transcriptome_5k_contigs_matches_reference(tmp_path):
    """Config 3's reference SHAPE: thousands of short contigs (more than the 64 the kernel keeps in LDS, three levels of the 64-ary contig
    search), 60 000 pairs, the CLI with the pairing rounds on the device against the reference binary, byte for byte.  (5 000 contigs, not
    100 000: the reference clears one std::set per contig and read, align.cpp:437-444.)"""
    import torch
    import synth_gpu
    tmp = str(tmp_path)
    p = B.Params("A:G", ["-M", "A:G"])
    G = synth_gpu.make_transcriptome(p, torch.device("cuda", 0), n_contigs=5000, seed=71)
    n = 60_000
    fa, f1, f2 = _pe_files(tmp, G, n, 150, 72)
    del G
    torch.cuda.empty_cache()
    flags = ["-M", "A:G", "-S", "1", "-s", "12", "-u", "-x", "700"]
    out = os.path.join(tmp, "out.sam")
    r = subprocess.run([BASAL_BIN, "-a", f1, "-b", f2, "-d", fa] + flags + ["-p", "16", "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    checker = orc.REF_BIN if os.path.exists(orc.REF_BIN) else orc.CLI
    c = subprocess.run([checker, "-a", "r1.fq", "-b", "r2.fq", "-d", "g.fa"] + flags + ["-p", "1", "-o", "ref.sam"], capture_output=True, text=True, cwd=tmp)
    assert c.returncode == 0, c.stderr
    got, want = sam_digest(out), sam_digest(os.path.join(tmp, "ref.sam"))
    assert got[1] == want[1] and got[1] >= 2 * n * 0.9
    assert got[0] == want[0]


def test_cli_pe_100k_contigs_device_pairing_equals_host_replay(tmp_path):
    """The two implementations of PairAlign's rounds -- on the device (basal_pe.hip) and replayed on the host over the mode-tagged logs
    (basal_pairs.cpp, BASAL_PE_HOST_PAIRING=1) -- on a 100 000-contig transcriptome stand-in, 100 000 pairs, -r 2: the same SAM."""
    import torch
    import synth_gpu
    tmp = str(tmp_path)
    p = B.Params("A:G", ["-M", "A:G"])
    G = synth_gpu.make_transcriptome(p, torch.device("cuda", 0), n_contigs=100_000, seed=81)
    n = 100_000
    fa, f1, f2 = _pe_files(tmp, G, n, 150, 82)
    del G
    torch.cuda.empty_cache()
    flags = ["-M", "A:G", "-S", "1", "-u", "-x", "700", "-r", "2"]
    outs = []
    for env in ({}, {"BASAL_PE_HOST_PAIRING": "1"}):
        out = os.path.join(tmp, "out%d.sam" % len(outs))
        r = subprocess.run([BASAL_BIN, "-a", f1, "-b", f2, "-d", fa] + flags + ["-p", "16", "-o", out], capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        outs.append(sam_digest(out))
    assert outs[0][1] >= 2 * n * 0.9
    assert outs[0] == outs[1]


def test_reference_host_worker_threads_share_one_core(tmp_path):
    """The unmodified reference host on the core with -p 8 (oracle/_ref_gpu/basal): eight SingleAlign objects call basal_core_align_batch on
    ONE core at the same time (each call takes a lane of the core: its own stream, staging and hit logs).  600 000 dirty, variable-length
    reads = 12 batches of the reference's 50 000: with -p 8 the batches come out in any order and a read's inherited state depends on which
    worker had the batch before (as in the reference itself), so the comparison is per read name against -p 1 of the same binary for every
    read whose seeds cannot inherit anything ((len - I + 1) % k != 0), and the record count for all."""
    ref_gpu = os.path.join(H.ROOT, "oracle", "_ref_gpu", "basal")
    if not os.path.exists(ref_gpu):
        pytest.skip("oracle/_ref_gpu/basal not built (tools/build_ref_with_core.sh needs /root/reference)")
    fa, fq = make_files(str(tmp_path), "C:T", 600_000, 100, seed=61, min_len=60)
    outs = {}
    for threads in ("1", "8"):
        r = subprocess.run([ref_gpu, "-a", "r.fq", "-d", "g.fa", "-M", "C:T", "-S", "1", "-s", "12", "-u", "-p", threads, "-o", "o%s.sam" % threads], capture_output=True, text=True,
                           cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr
        recs = {}
        for line in open(tmp_path / ("o%s.sam" % threads)):
            if not line.startswith("@"):
                recs[line.split("\t", 1)[0]] = line
        outs[threads] = recs
    assert len(outs["1"]) == len(outs["8"]) == 600_000
    same = plain = 0
    for name, line in outs["1"].items():
        seqlen = len(line.split("\t")[9])
        if (seqlen - 4 + 1) % 12 != 0:  # -I 4, -s 12: this read computes its own start offset
            plain += 1
            same += outs["8"][name] == line
    assert plain > 400_000 and same == plain
