"""The multi-GPU layers on the one GPU a test box has.
* basal_pipe_create_multi (`basal -G a,b,...`, single-end): whole batches fan out over one core per GPU, the carry state follows the batch
  numbers from GPU to GPU.  A GPU may be listed twice (two cores on one GPU), so the hand-over between ranks really runs here: `-G 0,0`.
* basal_multi_* (paired-end with -G; reads sharded by read number, RCCL gather of the hit records to GPU 0): one rank, so the collective
  degenerates, but every call of the sharded path runs -- ncclCommInitAll, the grouped ncclGather of records and hit streams, the
  stream-offset fix-up.  The sharding rule itself is covered for world sizes > 1 in tests/test_dist_gloo.py (CPU, same C function); more
  ranks than GPUs cannot be rehearsed for this layer: RCCL refuses a device listed twice."""
import os
import subprocess

import numpy as np
import pytest

import basal_amd as B
import harness as H
import oracle as orc

pytestmark = pytest.mark.gpu
BASAL_BIN = os.path.join(H.ROOT, "basal_amd", "bin", "basal")


@pytest.mark.parametrize("name,mode", [("ct_basic", B.STREAM_NONE), ("rep_r2_w10", B.STREAM_BEST), ("varlen_trim", B.STREAM_ALL), ("acgt_g2", B.STREAM_ALL)])
def test_multi_align_batch_equals_core_align_batch(name, mode):
    fa, fq, _, _ = H.fixture_paths(name)
    flags = H.MANIFEST[name]["flags"]
    p = B.Params(H.rule_of(flags), flags)
    ref = B.Reference(p, fasta_path=fa)
    ref.build_index(4)
    core = B.Core(p)
    core.upload(ref)
    multi = B.Multi(p, [0])
    multi.upload(ref)
    recs = H.filter_reads(p, orc.read_fastx(fq))
    bases, descs, stales = H.make_batch(p, recs)
    want, wstream, wc = core.align_batch(bases, descs, mode, stream_cap=100000, stales=stales)
    got, gstream, gc = multi.align_batch(bases, descs, mode, stream_cap=100000, stales=stales)
    assert np.array_equal(gc, wc)
    for f in ("best_level", "n_hit", "n_chit", "status", "stream_n"):
        assert np.array_equal(got[f], want[f]), f
    hit = want["best_level"] != 0xFF
    assert got["best"][hit].tobytes() == want["best"][hit].tobytes()
    # each GPU is sent what its shard needs and no more: with one rank that is every read's bases once, the descriptors and the stale entries in use
    live = descs["len"] > 0
    used = descs["stale_idx"][live & (descs["stale_idx"] != B.STALE_NONE)]
    span = int((descs["seq_off"][live].astype(np.int64) + descs["len"][live]).max() - descs["seq_off"][live].min())
    want_bytes = span + len(descs) * 16 + (int(used.max() - used.min() + 1) * 124 if len(used) else 0)
    assert B.lib().basal_multi_last_h2d_bytes(multi.h, 0) == want_bytes
    if mode != B.STREAM_NONE:  # the same records per read (stream positions may differ)
        for g, w in zip(got, want):
            a = gstream[g["stream_first"]: g["stream_first"] + g["stream_n"]].tobytes()
            b = wstream[w["stream_first"]: w["stream_first"] + w["stream_n"]].tobytes()
            assert a == b


@pytest.mark.parametrize("name", ["ct_basic", "varlen_trim", "rep_r2_w10", "tdel_pipeline", "pe_ct_100_u", "pe_rep_r2"])
def test_cli_sharded_path_matches_golden(name, tmp_path):
    """With BASAL_MULTI_HOST `basal -G 0,1,...` takes the sharded path (host QC and SAM text, basal_multi_align_batch, one RCCL gather per
    batch); BASAL_FORCE_MULTI runs it on one GPU."""
    fa, fq, fq2, _ = H.fixture_paths(name)
    pe = H.MANIFEST[name]["pe"]
    out = tmp_path / "o.sam"
    r = subprocess.run([BASAL_BIN, "-a", fq] + (["-b", fq2] if pe else []) + ["-d", fa] + H.MANIFEST[name]["flags"] + ["-p", "4", "-G", "0", "-Z", "300", "-o", str(out)],
                       capture_output=True, text=True, env=dict(os.environ, BASAL_FORCE_MULTI="1", BASAL_MULTI_HOST="1"))
    assert r.returncode == 0, r.stderr
    got = "".join(l for l in open(out) if not l.startswith("@PG"))
    assert got == H.golden_sam(name)


SE_TWO_RANKS = [n for n in H.SE if n.startswith(("varlen", "ct_basic", "rep_r2", "tdel", "acgt_g2", "long_", "dirty", "fasta_reads", "contigs_5k"))]


@pytest.mark.parametrize("gpus,pipe_bytes", [("0,0", "8192"), ("0,0,0", "20000"), ("0,0", None)])
@pytest.mark.parametrize("name", SE_TWO_RANKS)
def test_cli_two_ranks_one_pipeline_matches_golden(name, gpus, pipe_bytes, tmp_path):
    """`basal -G 0,0`: two (three) cores behind one batch pipeline, batches alternating between them -- with small batches the state a
    SingleAlign carries from read to read (the varlen_* fixtures' inherited offsets and seed slots) crosses from rank to rank hundreds of
    times, and the irregular-text fallback (fasta_reads, dirty reads) rewinds a pipe whose batches are spread over the ranks."""
    assert len(SE_TWO_RANKS) >= 8
    fa, fq, _, _ = H.fixture_paths(name)
    out = tmp_path / "o.sam"
    env = dict(os.environ)
    if pipe_bytes:  # small batches of the uncompressed file: the GPU finds the records in the text (the .gz form is parsed on the host)
        env["BASAL_PIPE_BYTES"] = pipe_bytes
        import gzip
        plain = tmp_path / os.path.basename(fq)[:-3]
        plain.write_bytes(gzip.open(fq, "rb").read())
        fq = str(plain)
    r = subprocess.run([BASAL_BIN, "-a", fq, "-d", fa] + H.MANIFEST[name]["flags"] + ["-p", "4", "-G", gpus, "-o", str(out)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    got = "".join(l for l in open(out) if not l.startswith("@PG"))
    assert got == H.golden_sam(name)


@pytest.mark.parametrize("name", ["varlen_trim", "varlen_s16", "tdel_pipeline", "rep_r2_w10"])
@pytest.mark.parametrize("one_stream", ["0", "1"])
def test_cli_two_kernel_streams_match_golden(name, one_stream, tmp_path):
    """Round 4: consecutive batches' kernels alternate between two kernel streams of the GPU (the end of one align launch overlaps the next batch's
    kernels); what orders them is the carry state's event, as between GPUs.  Small batches of the fixtures whose reads inherit state across batches,
    with both streams (default) and with one (BASAL_PIPE_ONE_STREAM=1): the golden SAMs either way."""
    import gzip
    fa, fq, _, _ = H.fixture_paths(name)
    plain = tmp_path / os.path.basename(fq)[:-3]
    plain.write_bytes(gzip.open(fq, "rb").read())
    out = tmp_path / "o.sam"
    env = dict(os.environ, BASAL_PIPE_BYTES="6000")
    if one_stream == "1":
        env["BASAL_PIPE_ONE_STREAM"] = "1"
    r = subprocess.run([BASAL_BIN, "-a", str(plain), "-d", fa] + H.MANIFEST[name]["flags"] + ["-p", "4", "-o", str(out)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "".join(l for l in open(out) if not l.startswith("@PG")) == H.golden_sam(name)


@pytest.mark.parametrize("name", [n for n in H.MANIFEST if H.MANIFEST[n]["pe"]])
def test_cli_two_ranks_paired_end_matches_golden(name, tmp_path):
    """`basal -b ... -G 0,0`: the paired-end pipeline over two cores, 32 pairs per batch (the carry state of both mates' slots crosses from rank
    to rank with every batch), plain-text mate files."""
    import gzip
    fa, fq, fq2, _ = H.fixture_paths(name)
    f1, f2 = tmp_path / "r1.fq", tmp_path / "r2.fq"
    f1.write_bytes(gzip.open(fq, "rb").read())
    f2.write_bytes(gzip.open(fq2, "rb").read())
    out = tmp_path / "o.sam"
    r = subprocess.run([BASAL_BIN, "-a", str(f1), "-b", str(f2), "-d", fa] + H.MANIFEST[name]["flags"] + ["-p", "4", "-G", "0,0", "-Z", "64", "-o", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "".join(l for l in open(out) if not l.startswith("@PG")) == H.golden_sam(name)


def test_pipe_two_ranks_equals_one_rank():
    """basal_pipe_create_multi through the ABI: the same text batches through a one-core pipe and through a pipe over two cores (same GPU
    listed twice) give the same SAM bytes, batch by batch."""
    name = "varlen_trim"
    fa, fq, _, _ = H.fixture_paths(name)
    flags = H.MANIFEST[name]["flags"]
    p = B.Params(H.rule_of(flags), flags)
    ref = B.Reference(p, fasta_path=fa)
    ref.build_index(4)
    cores = [B.Core(p), B.Core(p)]
    for c in cores:
        c.upload(ref)
        c.set_contig_names(ref.names())
    text = open(fq, "rb").read() if not fq.endswith(".gz") else __import__("gzip").open(fq, "rb").read()
    lines = text.rstrip(b"\n").split(b"\n")
    per = 4 * 23  # 23 reads per batch
    batches = [b"\n".join(lines[i:i + per]) + b"\n" for i in range(0, len(lines), per)]
    outs = []
    for cs in (cores[0], cores):
        pipe = B.Pipe(cs, depth=2, max_reads=4096, max_bytes=1 << 20)
        got, inflight = [], 0
        for b in batches:
            if inflight == (2 if cs is cores[0] else 4):
                rc, data, _ = pipe.collect()
                assert rc == 0, data
                got.append(data)
                inflight -= 1
            pipe.submit_text(b)
            inflight += 1
        while inflight:
            rc, data, _ = pipe.collect()
            assert rc == 0, data
            got.append(data)
            inflight -= 1
        pipe.close()
        outs.append(got)
    assert len(outs[0]) == len(batches) and outs[0] == outs[1]


def test_multi_refuses_a_device_twice():
    p = B.Params("C:T", ["-M", "C:T"])
    with pytest.raises(B.BasalError):
        B.Multi(p, [0, 0])
