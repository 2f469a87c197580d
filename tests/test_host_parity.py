"""Product host code (libbasal_amd.so, no GPU call) against the oracle: -M code tables, the packed
reference, blocks, the seed index and its cut-off, FilterReads, and the exported ABI."""
import ctypes as C
import re
import os

import numpy as np
import pytest

import basal_amd as B
from basal_amd import core as bc
import harness as H
import oracle as orc

RULES = ["C:T", "A:G", "A:CGT", "T:-", "G:ACT-", "T:C", "G:A", "c:t", "A:GG", "C:TA-"]


@pytest.mark.parametrize("rule", RULES)
def test_set_align_tables(rule):
    p = B.Params(rule)
    o = orc.make_param(["-M", rule])
    for f in ("alphabet", "rev_alphabet", "reg_alphabet", "alphabet_mread", "rev_alphabet_mread"):
        assert bytes(getattr(p.c, f)) == bytes(getattr(o, f)), f
    assert p.c.useful_nt[:8] == o.useful_nt[:8]
    assert p.c.new_rule == o.new_rule
    assert p.c.readnt_cnt == o.readnt_cnt


def test_known_code_tables():
    """C:T -> A0 C1 G2 T3; A:G -> A1 C0 G3 T2; A:CGT -> A1 C0 G2 T3; T:- -> T1 A0 C2 G3."""
    want = {"C:T": dict(A=0, C=1, G=2, T=3), "A:G": dict(A=1, C=0, G=3, T=2), "A:CGT": dict(A=1, C=0, G=2, T=3),
            "T:-": dict(T=1, A=0, C=2, G=3)}
    for rule, tab in want.items():
        p = B.Params(rule)
        for b, code in tab.items():
            assert p.c.alphabet[ord(b)] == code and p.c.alphabet[ord(b.lower())] == code


@pytest.mark.parametrize("bad", ["CT", "N:T", "C:C", "C:X", ""])
def test_set_align_rejects(bad):
    with pytest.raises(B.BasalError):
        B.Params(bad)


def test_param_flag_order_min_read_size():
    # -s sets min_read_size = k + I - 1 with I as parsed so far (param.cpp:112); default stays 16 (param.cpp:34)
    assert B.Params("C:T").c.min_read_size == 16
    assert B.Params("C:T", ["-s", "12"]).c.min_read_size == 15
    assert B.Params("C:T", ["-I", "2", "-s", "12"]).c.min_read_size == 13
    assert B.Params("C:T", ["-s", "12", "-I", "2"]).c.min_read_size == 15
    for fl in ([], ["-s", "12"], ["-I", "2", "-s", "12"], ["-s", "12", "-I", "2"]):
        assert B.Params("C:T", fl).c.min_read_size == orc.make_param(["-M", "C:T"] + fl).min_read_size


@pytest.mark.parametrize("name", ["ct_n1_dirty", "tx_ag_150", "rep_r1", "v_frac05_I2", "acgt_g2"])
def test_reference_and_index_match_oracle(name):
    fa = H.fixture_paths(name)[0]
    flags = H.MANIFEST[name]["flags"]
    p = B.Params(H.rule_of(flags), flags)
    ref = B.Reference(p, fasta_path=fa)
    o = orc.Oracle(flags, fa)
    a = o.arrays()
    assert ref.names() == a["names"]
    assert np.array_equal(ref.sizes(), a["size"])
    assert np.array_equal(ref.rc_offsets(), a["rc_offset"])
    assert np.array_equal(ref.anchors(), a["anchor"])
    assert np.array_equal(ref.words(0), a["xref0"])
    assert np.array_equal(ref.words(1), a["xref1"])
    for threads in (1, 3):
        ref.build_index(threads)
        off, nfwd, locs, mk = ref.index()
        assert np.array_equal(off.astype(np.uint64), a["off"])
        assert np.array_equal(nfwd, a["n_fwd"])
        assert np.array_equal(locs, a["locs"])
        assert mk == a["max_kmer_num"]
    o.close()


def test_default_seed_size_index_cutoff():
    """k=16: the cut-off index is computed in single precision (refbase.cpp:363) -> 43046699."""
    name = "c1_s16"
    fa = H.fixture_paths(name)[0]
    flags = H.MANIFEST[name]["flags"]
    p = B.Params("C:T", flags)
    ref = B.Reference(p, fasta_path=fa)
    ref.build_index(4)
    o = orc.Oracle(flags, fa)
    a = o.arrays()
    off, nfwd, locs, mk = ref.index()
    assert mk == a["max_kmer_num"]
    assert np.array_equal(locs, a["locs"]) and np.array_equal(nfwd, a["n_fwd"])
    o.close()


@pytest.mark.parametrize("name,extra", [("varlen_trim", []), ("ct_n1_dirty", []), ("varlen_s16", ["-q", "30"]),
                                         ("ct_basic", ["-A", "AGATCGGAAGAGC", "-q", "20", "-z", "64"])])
def test_filter_reads_matches_oracle(name, extra):
    fa, fq, _, _ = H.fixture_paths(name)
    flags = H.MANIFEST[name]["flags"] + extra
    p = B.Params(H.rule_of(flags), flags)
    o = orc.Oracle([f for f in flags], fa)
    reads = orc.read_fastx(fq)
    # make the extra-flag cases bite: low-quality tails and adapter read-through
    if extra:
        reads = [(n, s[:60] + "AGATCGGAAGAGCACACG"[: max(0, len(s) - 60)] if i % 3 == 0 else s,
                  (q[:-15] + "#" * 15) if i % 2 == 0 else q) for i, (n, s, q) in enumerate(reads)]
        reads = [(n, s, q[: len(s)].ljust(len(s), "I")) for n, s, q in reads]
    mine = H.filter_reads(p, reads)
    for i, ((n, s, q), m) in enumerate(zip(reads, mine)):
        s2, q2 = s[: p.c.max_readlen], q[: p.c.max_readlen]
        ref = o.align(i, 0, n, s2, q2)
        assert bool(m["qc"]) == ref["filtered"], (i, n)
        assert m["seq"] == ref["seq"] and m["qual"] == ref["qual"], (i, n)
        if not ref["filtered"]:
            assert m["max_snp"] == ref["max_snp"], (i, n)
    o.close()


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(H.ROOT, "include", "basal_core.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(basal_(?:core|host|last|pipe|multi|shard)_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 35
    L = C.CDLL(B.lib_path())
    for s in sorted(declared):
        assert hasattr(L, s), "libbasal_amd.so does not export " + s
    assert declared == {n for n, _, _ in bc.SYMBOLS}


def test_core_fails_loudly_without_gpu():
    """No CPU fallback: without a HIP device creating the core is an error, not a silent slow path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(B.BasalError):
        B.Core(B.Params("C:T"))


def test_transcriptome_generator_layout():
    """tools/synth_gpu.make_transcriptome (bench.py --config 3's many-contig reference) lays contigs out as RefSeq does: each contig on a
    word boundary with two pad words, forward strand from the slot's start, reverse complement ending at the slot's end (refbase.cpp:222-244)."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(H.ROOT, "tools"))
    import synth_gpu
    p = B.Params("A:G", ["-M", "A:G"])
    G = synth_gpu.make_transcriptome(p, torch.device("cpu"), n_contigs=300, seed=5)
    al = [p.c.alphabet[ord(b)] for b in "ACGT"]
    rv = [p.c.rev_alphabet[ord(b)] for b in "ACGT"]
    fw = G.words[0].numpy().view(np.uint64)
    rc = G.words[1].numpy().view(np.uint64)
    ids = G.ids.numpy()
    off = G.base_off.numpy()

    def base(words, gpos):
        return int(words[gpos >> 5] >> np.uint64(62 - 2 * (gpos & 31))) & 3
    rng = np.random.default_rng(1)
    for c in [0, 1, 299] + list(rng.integers(0, 300, 20)):
        L, a = G.sizes[c], int(G.anchors[c])
        assert int(G.anchors[c + 1]) - a == int(G.rc_offsets[c]) == ((L + 31) // 32 + 2) * 32
        for i in [0, 1, 31, 32, L // 2, L - 1]:
            assert base(fw, a + i) == al[ids[off[c] + i]]
            assert base(rc, a + int(G.rc_offsets[c]) - 1 - i) == rv[ids[off[c] + i]]
        assert base(fw, a + L) == 0 and base(rc, a + int(G.rc_offsets[c]) - L - 1) == 0
    blk = G.blocks
    assert blk.shape == (600, 3) and (blk[0] == [0, 0, G.sizes[0]]).all() and (blk[1] == [1, int(G.rc_offsets[0]) - G.sizes[0], int(G.rc_offsets[0])]).all()
    m1, m2 = synth_gpu.make_pairs(G, 50, torch.device("cpu"), read_len=150, seed=3, p_conv=0.0, sub_rate=0.0, rev_frac=0.0)
    flat = "".join("ACGT"[x] for x in ids)
    comp = str.maketrans("ACGT", "TGCA")
    for k in range(50):
        r1 = bytes(m1[k * 150:(k + 1) * 150].numpy()).decode()
        r2 = bytes(m2[k * 150:(k + 1) * 150].numpy()).decode()
        s = flat.find(r1)
        assert s >= 0
        e = flat.find(r2.translate(comp)[::-1])
        assert e >= s and e + 150 - s <= 600  # mate 2 = reverse complement of the fragment's end, same contig


@pytest.mark.parametrize("layout", ["regular", "crlf_and_wrapped", "gt_in_sequence_line"])
def test_large_fasta_loader_matches_oracle(layout, tmp_path):
    """FASTA files of a megabyte and more take basal_host_ref_load's parallel path (records found at their line starts, sequences gathered
    and packed by several threads) when they are laid out the usual way, and the token reader otherwise: either way the packed strands,
    anchors and unmasked blocks are the oracle's (which reads the file the way RefSeq::LoadNextSeq does)."""
    rng = np.random.default_rng(17)
    parts = []
    for c in range(7):
        n = int(rng.integers(150_000, 600_000))
        s = rng.choice(np.frombuffer(b"ACGT", np.uint8), n)
        for _ in range(5):  # N runs, lower case, other IUPAC letters
            a = int(rng.integers(0, n - 3000))
            s[a:a + int(rng.integers(1, 2500))] = ord("N")
        lo = int(rng.integers(0, n - 5000))
        s[lo:lo + 4000] |= 0x20
        s[rng.integers(0, n, 20)] = ord("R")
        w = [60, 70, 61, 80, 50, 100, 64][c]
        body = b"\n".join(s[i:i + w].tobytes() for i in range(0, n, w))
        parts.append(b">ctg%d some description > with an arrow\n" % c + body + b"\n")
    text = b"".join(parts)
    if layout == "crlf_and_wrapped":
        text = text.replace(b"\n", b"\r\n", 3000)            # a stretch of CRLF lines: white space inside what a line reader sees
    elif layout == "gt_in_sequence_line":
        k = text.index(b"\n", 2_000_000)
        text = text[:k] + b" >x\nACGTACGTACGTACGTACGTACGT" + text[k:]  # a token that starts a record in the middle of a line
    fa = str(tmp_path / "big.fa")
    open(fa, "wb").write(text)
    flags = ["-M", "C:T", "-s", "12"]
    p = B.Params("C:T", flags)
    ref = B.Reference(p, fasta_path=fa)
    o = orc.Oracle(flags, fa)
    a = o.arrays()
    assert ref.names() == a["names"]
    assert np.array_equal(ref.sizes(), a["size"])
    assert np.array_equal(ref.anchors(), a["anchor"])
    assert np.array_equal(ref.words(0), a["xref0"])
    assert np.array_equal(ref.words(1), a["xref1"])
    ref.build_index(4)
    off, nfwd, locs, mk = ref.index()
    assert np.array_equal(locs, a["locs"]) and np.array_equal(nfwd, a["n_fwd"]) and mk == a["max_kmer_num"]
    o.close()
