"""The N>1 path on CPU: world_size-2 gloo processes shard reads by global read number, align their
shard (the oracle stands in for the GPU core here) and gather fixed-size records to rank 0. The merged
result must equal a single-process run -- in particular the myrand-driven choices must not depend on
how the reads were split."""
import os
import sys

import numpy as np
import pytest

import harness as H

REC = np.dtype([("best_level", "<u4"), ("n_hit", "<u4"), ("n_chit", "<u4"), ("chr", "<u4"), ("loc", "<u4"), ("gap_size", "<i4"),
                ("gap_pos", "<u4"), ("chain", "<u4")])


def _align_range(name, b, e):
    import oracle as orc
    fa, fq, _, _ = H.fixture_paths(name)
    flags = H.MANIFEST[name]["flags"]
    o = orc.Oracle(flags, fa)
    reads = orc.read_fastx(fq)
    out = np.zeros(e - b, REC)
    for i in range(b, e):
        n, s, q = reads[i]
        r = o.align(i, 0, n, s, q)  # i = GLOBAL read number
        rec = out[i - b]
        rec["best_level"] = 0xFF
        if r["filtered"] or not r["log"]:
            continue
        lvl = min(h[5] for h in r["log"])
        hits = [h for h in r["log"] if h[5] == lvl and h[6] == 0] + [h for h in r["log"] if h[5] == lvl and h[6] == 1]
        j = 0 if len(hits) == 1 else orc.lib().orc_myrand(i, o.p.randseed) % len(hits)
        h = hits[j]
        rec["best_level"], rec["n_hit"], rec["n_chit"] = lvl, sum(1 for x in hits if x[6] == 0), sum(1 for x in hits if x[6] == 1)
        rec["chr"], rec["loc"], rec["gap_size"], rec["gap_pos"], rec["chain"] = h[1], h[0], h[2], h[4], h[6]
    o.close()
    return out


def _worker(rank, world, port, name, n, q):
    import torch.distributed as dist
    sys.path.insert(0, H.ROOT)
    from basal_amd import dist as bd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    merged = bd.align_sharded(lambda b, e: _align_range(name, b, e), n, rank, world, dist)
    if rank == 0:
        q.put(merged.tobytes())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,n", [("rep_r1", 101), ("ct_basic", 64)])
def test_two_rank_gloo_matches_single_process(name, n):
    import torch.multiprocessing as mp
    from basal_amd import dist as bd
    assert [bd.shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert [bd.shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = np.frombuffer(q.get(timeout=300), dtype=REC)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = _align_range(name, 0, n)
    assert len(merged) == n
    assert merged.tobytes() == single.tobytes()
