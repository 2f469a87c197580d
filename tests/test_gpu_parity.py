"""Parity tests proper: the HIP path (through the C ABI) against the oracle and the golden SAMs."""
import os
import subprocess

import numpy as np
import pytest

import basal_amd as B
import harness as H
import oracle as orc

pytestmark = pytest.mark.gpu
BASAL_BIN = os.path.join(H.ROOT, "basal_amd", "bin", "basal")


def run_product(name, stream_mode, extra_flags=(), batch=None):
    fa, fq, _, _ = H.fixture_paths(name)
    flags = H.MANIFEST[name]["flags"] + list(extra_flags)
    p = B.Params(H.rule_of(flags), flags)
    ref = B.Reference(p, fasta_path=fa)
    ref.build_index(4)
    core = B.Core(p)
    core.upload(ref)
    reads = orc.read_fastx(fq)
    recs = H.filter_reads(p, reads)
    results, streams = [], []
    carry = None
    tracker = H.StaleTracker(p)
    step = batch or len(recs)
    all_stream = []
    for b0 in range(0, len(recs), step):
        part = recs[b0:b0 + step]
        bases, descs, stales = H.make_batch(p, part, tracker)
        res, stream, carry = core.align_batch(bases, descs, stream_mode, stream_cap=200000, carry=carry, stales=stales)
        res = res.copy()
        res["stream_first"] += len(all_stream)
        results.append(res)
        all_stream.extend(stream)
    results = np.concatenate(results)
    stream = np.array(all_stream, dtype=B.core.HIT_DTYPE) if all_stream else np.zeros(0, B.core.HIT_DTYPE)
    return p, ref, core, recs, results, stream


@pytest.mark.parametrize("name", H.SE)
def test_hit_logs_match_oracle(name):
    """Every stored hit of every read, in insertion order, with level/chain/mode: the whole AddHit history."""
    fa, fq, _, _ = H.fixture_paths(name)
    flags = H.MANIFEST[name]["flags"]
    p, ref, core, recs, results, stream = run_product(name, B.STREAM_ALL)
    o = orc.Oracle(flags, fa)
    reads = orc.read_fastx(fq)
    nbad = 0
    for i, ((n, s, q), r, res) in enumerate(zip(reads, recs, results)):
        exp = o.align(i, 0, n, s[: p.c.max_readlen], q[: p.c.max_readlen])
        assert bool(r["qc"]) == exp["filtered"]
        if exp["filtered"]:
            assert res["status"] == 1
            continue
        got = [H.hit_tuple(h) for h in stream[res["stream_first"]: res["stream_first"] + res["stream_n"]]] if res["best_level"] != 0xFF else []
        if got != exp["log"]:
            nbad += 1
            if nbad <= 3:
                print("MISMATCH read", i, n, "\n got", got[:6], "\n exp", exp["log"][:6])
        assert tuple(res["start_off"][:1]) == exp["start_off"][:1] or p.c.chains == 2
    assert nbad == 0
    o.close()


@pytest.mark.parametrize("name", H.SE)
def test_sam_matches_golden_through_abi(name):
    flags = H.MANIFEST[name]["flags"]
    r2 = "-r" in flags and flags[flags.index("-r") + 1] == "2"
    p, ref, core, recs, results, stream = run_product(name, B.STREAM_BEST if r2 else B.STREAM_NONE)
    body = "".join(H.format_se(p, ref, recs, results, stream))
    assert H.sam_header(ref) + body == H.golden_sam(name)


@pytest.mark.parametrize("name", ["varlen_trim", "rep_r2_w10", "ct_g3"])
def test_small_batches_carry_state(name):
    """Splitting the input into small batches must not change anything (carry of the stale start offset)."""
    flags = H.MANIFEST[name]["flags"]
    r2 = "-r" in flags and flags[flags.index("-r") + 1] == "2"
    p, ref, core, recs, results, stream = run_product(name, B.STREAM_BEST if r2 else B.STREAM_NONE, batch=37)
    body = "".join(H.format_se(p, ref, recs, results, stream))
    assert H.sam_header(ref) + body == H.golden_sam(name)


@pytest.mark.parametrize("name", H.SE)
def test_cli_sam_matches_golden(name, tmp_path):
    fa, fq, _, _ = H.fixture_paths(name)
    out = tmp_path / "o.sam"
    env = dict(os.environ, BASAL_CPU_INDEX="1")
    r = subprocess.run([BASAL_BIN, "-a", fq, "-d", fa] + H.MANIFEST[name]["flags"] + ["-p", "4", "-o", str(out)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    got = "".join(l for l in open(out) if not l.startswith("@PG"))
    assert got == H.golden_sam(name)


@pytest.mark.parametrize("name", ["ct_n1_dirty", "tx_ag_150", "rep_r1", "v_frac05_I2", "c1_s16"])
def test_gpu_index_build_matches_cpu_build(name):
    """basal_core_build_index (counting + stable radix sort on the GPU) == the host build == the oracle."""
    fa = H.fixture_paths(name)[0]
    flags = H.MANIFEST[name]["flags"]
    p = B.Params(H.rule_of(flags), flags)
    ref = B.Reference(p, fasta_path=fa)
    ref.build_index(4)
    off, nfwd, locs, mk = ref.index()
    core = B.Core(p)
    mk_gpu = core.upload(ref, build_on_gpu=True)
    goff, gnfwd, glocs, gmk = core.get_index(len(nfwd))
    assert mk_gpu == mk == gmk
    assert np.array_equal(goff, off)
    assert np.array_equal(gnfwd, nfwd)
    assert np.array_equal(glocs, locs)


def test_cli_with_gpu_index(tmp_path):
    name = "acgt_g2"
    fa, fq, _, _ = H.fixture_paths(name)
    out = tmp_path / "o.sam"
    env = {k: v for k, v in os.environ.items() if k != "BASAL_CPU_INDEX"}
    r = subprocess.run([BASAL_BIN, "-a", fq, "-d", fa] + H.MANIFEST[name]["flags"] + ["-p", "2", "-o", str(out)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "seed table (GPU)" in r.stderr
    got = "".join(l for l in open(out) if not l.startswith("@PG"))
    assert got == H.golden_sam(name)


@pytest.mark.parametrize("name", H.PE)
def test_cli_pe_sam_matches_golden(name, tmp_path):
    """Paired-end: both mates aligned on the GPU (all modes, mode-tagged logs), pairing rounds on the host."""
    fa, fq, fq2, _ = H.fixture_paths(name)
    out = tmp_path / "o.sam"
    r = subprocess.run([BASAL_BIN, "-a", fq, "-b", fq2, "-d", fa] + H.MANIFEST[name]["flags"] + ["-p", "3", "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = "".join(l for l in open(out) if not l.startswith("@PG"))
    exp = H.golden_sam(name)
    if got != exp:
        g, e = got.splitlines(), exp.splitlines()
        for i, (x, y) in enumerate(zip(g, e)):
            if x != y:
                print("first difference at line", i, "\n got", x[:300], "\n exp", y[:300])
                break
    assert got == exp


@pytest.mark.parametrize("name", ["pe_rep_r2", "pe_dirty_r1"])
def test_cli_pe_small_batches(name, tmp_path):
    fa, fq, fq2, _ = H.fixture_paths(name)
    out = tmp_path / "o.sam"
    r = subprocess.run([BASAL_BIN, "-a", fq, "-b", fq2, "-d", fa] + H.MANIFEST[name]["flags"] + ["-Z", "50", "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "".join(l for l in open(out) if not l.startswith("@PG")) == H.golden_sam(name)
