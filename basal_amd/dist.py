"""Read-sharded multi-GPU driver: one process per GPU, reference + index replicated, reads split into
contiguous ranges by global read number, one gather of the fixed-size results to rank 0.

torch.distributed is plumbing only (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in CPU tests).
The per-shard aligner is a callable so the same code drives basal_core_align_batch* on a GPU and,
in the CPU tests, the oracle.
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous slice [begin, end) of n reads for this rank; slices differ by at most one read. This IS the native rule: the C
    function basal_shard_range of libbasal_amd.so, the one basal_multi_align_batch (the `basal -G 0,1,...` path) shards with."""
    from .core import shard_range as native
    return native(n, rank, world)


def gather_results(local, n_total, rank, world, dist, device=None):
    """local: np.ndarray of fixed-size records for this rank's shard (in shard order).
    Returns the n_total records in global read order on rank 0, None elsewhere. One collective."""
    import torch
    itemsize = local.dtype.itemsize
    counts = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    mx = max(counts)
    buf = torch.zeros(mx * itemsize, dtype=torch.uint8, device=device)
    if len(local):
        buf[: len(local) * itemsize] = torch.from_numpy(np.ascontiguousarray(local).view(np.uint8).reshape(-1)).to(buf.device)
    out = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    parts = [np.frombuffer(o.cpu().numpy().tobytes(), dtype=local.dtype)[: counts[r]] for r, o in enumerate(out)]
    return np.concatenate(parts)


def align_sharded(align_shard, n_total, rank, world, dist, device=None):
    """align_shard(begin, end) -> records for reads [begin, end) (global read numbers feed myrand, so a
    shard must be aligned with its global indices). Returns all records on rank 0."""
    b, e = shard_range(n_total, rank, world)
    return gather_results(align_shard(b, e), n_total, rank, world, dist, device)
