"""bench.py's distributed branch rehearsed on the one GPU a test box has (the driver's SCALE run is the only place N > 1 GPUs exist):
the process group, the per-step gather of the records to rank 0 and the line's N > 1 fields, with the NCCL (= RCCL) backend in one rank and
with two gloo ranks sharing the GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

import harness as H

pytestmark = pytest.mark.gpu
BENCH = os.path.join(H.ROOT, "bench.py")
SMALL = ["--genome-scale", "0.02", "--batch", "200000", "--steps", "2", "--warmup", "1", "--ref-sample", "0"]


def free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def last_json(stdout):
    return json.loads(stdout.strip().splitlines()[-1])


def test_bench_distributed_branch_one_rank_rccl():
    """BASAL_BENCH_FORCE_DIST=1: init_process_group("nccl"), every step's records gathered with dist.gather (RCCL) behind the kernel, the
    barrier + max-over-ranks timing -- in one rank.  What rank 0 received is what the ranks aligned."""
    env = dict(os.environ, BASAL_BENCH_FORCE_DIST="1", BASAL_BENCH_NO_H2H="1", BASAL_BENCH_NO_UNIFORM="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    r = subprocess.run([sys.executable, BENCH] + SMALL + ["--cpu-sample", "20000"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    c = d["config"]
    assert c["gathered_aligned_reads"] is not None and c["gathered_aligned_reads"] == c["aligned_reads_all_ranks"] > 0.9 * 2 * 200000
    assert d["roofline"]["frac"] > 0 and "identical to the oracle" in d["cpu_baseline"]["sample"]


def test_bench_two_gloo_ranks_on_one_gpu():
    """The launch line the driver uses for N > 1 (torch.distributed.run, one process per rank) with BASAL_DIST_BACKEND=gloo and both ranks on GPU 0:
    each rank aligns its own reads, rank 0 gathers every step's records of both, the line says n_gpus 2 and counts both ranks' reads."""
    env = dict(os.environ, BASAL_DIST_BACKEND="gloo", BASAL_BENCH_NO_H2H="1", BASAL_BENCH_NO_UNIFORM="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                        BENCH, "--gpus", "2"] + SMALL, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
    c = d["config"]
    assert c["gathered_aligned_reads"] == c["aligned_reads_all_ranks"] > 0.9 * 2 * 2 * 200000
    assert d["roofline"]["kernel_ms"] > 0 and d["cpu_baseline"] is None


def test_bench_placement_draws_on_a_heavy_index():
    """bench.py's set-up on an index that keeps long lists (hg38-like genome at scale 0.4: the cut-off passes 32 768 by itself): up to four placements of the
    index timed (basal_core_placement_fork / _swap / _commit), the steps on the fastest, the sample still identical to the oracle."""
    env = dict(os.environ, BASAL_BENCH_NO_H2H="1", BASAL_BENCH_NO_UNIFORM="1")
    r = subprocess.run([sys.executable, BENCH, "--genome-scale", "0.4", "--batch", "1000000", "--steps", "2", "--warmup", "1", "--ref-sample", "0", "--cpu-sample", "20000"],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    draws = d["config"]["placement_draws_ms"]
    assert 2 <= len(draws) <= 4 and all(t > 0 for t in draws), draws
    assert "HEAVY" in d["roofline"]["kernel"] and "identical to the oracle" in d["cpu_baseline"]["sample"]
    # (the steps' launches -- other batches, copies running beside them -- stay near the best calibration launch, not the worst)
    assert d["roofline"]["kernel_ms"] < min(draws) * 1.10
