"""Compile-time guard (no GPU): the kernels' ISA must not contain the lane-split self-loop that a
miscompile of the persistent work loop produced, nor unresolved flat_ memory instructions, nor large
private arrays."""
import os
import subprocess
import sys

import harness as H


def test_isa_static_check():
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "check_isa.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ScratchSize" in r.stdout
    for line in r.stdout.splitlines():
        if "ScratchSize" in line:  # register spills only (the occupancy targets cost a few); no big private arrays
            assert int(line.split()[-1]) <= 512, "kernel uses a lot of private scratch memory: " + line
