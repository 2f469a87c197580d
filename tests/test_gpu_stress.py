"""Bounded stress of the batch pipeline over several cores on one GPU (`basal -G 0,0,0`): round 3 recorded ONE differing SAM record in about 600
executions of test_cli_two_ranks_one_pipeline_matches_golden[tdel_pipeline-0,0,0-20000] and the difference was not kept.  This runs the same
fixtures through a three-core pipe in one process (submitter and collector on their own threads, as in the command line), 20 000-byte batches,
a few hundred passes, and on any difference reports -- and leaves under gpurun_out/ -- the pass, the batch and both records.
(tools/stress_pipe.py and tools/stress_multi.py are the unbounded forms; profiles/r04_stress_multi.log holds round 4's 1 500 + 2 000 clean passes.)"""
import os
import sys

import pytest

import harness as H

sys.path.insert(0, os.path.join(H.ROOT, "tools"))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,iters", [("tdel_pipeline", 200), ("varlen_trim", 100), ("varlen_s16", 100)])
def test_three_cores_one_gpu_many_passes(name, iters):
    import stress_pipe
    log = os.path.join(H.ROOT, "gpurun_out", "stress_%s.log" % name)
    os.makedirs(os.path.dirname(log), exist_ok=True)
    bad, msgs = stress_pipe.stress(name, 3, 20000, iters, log=log, verbose=False)
    assert bad == 0, "%d of %d passes differ from the golden SAM (kept in %s):\n%s" % (bad, iters, log, "\n".join(msgs[:3]))


def test_cli_three_ranks_repeated(tmp_path):
    """The same through the command line (fresh processes: fresh allocations, whatever earlier processes left in HBM), 40 runs."""
    import subprocess
    log = os.path.join(H.ROOT, "gpurun_out", "stress_cli_tdel_pipeline.log")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "stress_multi.py"), "tdel_pipeline", "0,0,0", "20000", "40", "--log", log], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-500:]
