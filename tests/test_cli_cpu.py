"""The `basal` CLI without a GPU: argument handling mirrors the reference, and the hot path refuses to run
(no silent CPU fallback)."""
import os
import subprocess

import pytest

import harness as H

BASAL_BIN = os.path.join(H.ROOT, "basal_amd", "bin", "basal")


def run(args):
    return subprocess.run([BASAL_BIN] + args, capture_output=True, text=True)


def test_usage_and_required_flags():
    assert run([]).returncode != 0
    fa, fq, _, _ = H.fixture_paths("ct_basic")
    r = run(["-a", fq, "-d", fa])
    assert r.returncode != 0 and "-M option is required" in r.stderr
    r = run(["-a", fq, "-d", fa, "-M", "C:C"])
    assert r.returncode != 0 and "should not be equal" in r.stderr
    r = run(["-a", fq, "-d", fa, "-M", "C:T", "-s", "9"])
    assert r.returncode != 0 and "seed size" in r.stderr
    r = run(["-a", fq, "-d", fa, "-M", "C:T", "-Q", "1"])
    assert r.returncode != 0 and "unknown option" in r.stderr
    r = run(["-a", fq, "-d", "/nonexistent.fa", "-M", "C:T"])
    assert r.returncode != 0 and "reference file" in r.stderr


def test_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    fa, fq, _, _ = H.fixture_paths("ct_basic")
    r = run(["-a", fq, "-d", fa, "-M", "C:T", "-s", "12", "-o", str(tmp_path / "o.sam")])
    assert r.returncode != 0
    assert "GPU core" in r.stderr and "no CPU fallback" in r.stderr
