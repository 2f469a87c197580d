"""The multi-GPU layer (basal_multi_*: reads sharded by read number, RCCL gather of the hit records to GPU 0) on the one GPU a test box
has: one rank, so the collective degenerates, but every call of the sharded path runs -- ncclCommInitAll, the grouped ncclGather of
records and hit streams, the stream-offset fix-up.  The sharding rule itself is covered for world sizes > 1 in tests/test_dist_gloo.py
(CPU, same C function).  More ranks than GPUs cannot be rehearsed here: RCCL refuses a device listed twice."""
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
    if mode != B.STREAM_NONE:  # the same records per read (stream positions may differ)
        for g, w in zip(got, want):
            a = gstream[g["stream_first"]: g["stream_first"] + g["stream_n"]].tobytes()
            b = wstream[w["stream_first"]: w["stream_first"] + w["stream_n"]].tobytes()
            assert a == b


@pytest.mark.parametrize("name", ["ct_basic", "varlen_trim", "rep_r2_w10", "tdel_pipeline", "pe_ct_100_u", "pe_rep_r2"])
def test_cli_sharded_path_matches_golden(name, tmp_path):
    """`basal -G 0,1,...` takes the sharded path (host QC and SAM text, basal_multi_align_batch); BASAL_FORCE_MULTI runs it on one GPU."""
    fa, fq, fq2, _ = H.fixture_paths(name)
    pe = H.MANIFEST[name]["pe"]
    out = tmp_path / "o.sam"
    r = subprocess.run([BASAL_BIN, "-a", fq] + (["-b", fq2] if pe else []) + ["-d", fa] + H.MANIFEST[name]["flags"] + ["-p", "4", "-G", "0", "-Z", "300", "-o", str(out)],
                       capture_output=True, text=True, env=dict(os.environ, BASAL_FORCE_MULTI="1"))
    assert r.returncode == 0, r.stderr
    got = "".join(l for l in open(out) if not l.startswith("@PG"))
    assert got == H.golden_sam(name)


def test_multi_refuses_a_device_twice():
    p = B.Params("C:T", ["-M", "C:T"])
    with pytest.raises(B.BasalError):
        B.Multi(p, [0, 0])
