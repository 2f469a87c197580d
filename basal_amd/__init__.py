"""basal_amd -- MI355X-native seed-and-extend core for BASAL (ctypes view of libbasal_amd.so).

The product is the C-ABI library declared in include/basal_core.h (HIP kernels for gfx950 + the
C++ host helpers) and the `basal` command-line aligner built from it.  This package only mirrors
that ABI for tests and bench.py; it contains no alignment logic and no CPU fallback.
"""
from .core import (  # noqa: F401
    BasalError, Core, Params, Reference, lib, lib_path, build,
    basal_hit, basal_read, basal_result, basal_params, basal_stale, basal_mate, READ_ALLMODES,
    STREAM_NONE, STREAM_BEST, STREAM_ALL, STALE_NONE, STALE_CARRY,
    Multi, shard_range, Pipe, basal_rawread, basal_pipe_opts, basal_batch_stats, PIPE_OUT_SAM, PIPE_OUT_RESULTS, FMT_FASTQ, FMT_FASTA, RAWREAD_DTYPE,
)
