"""The CPU oracle against the golden SAMs produced by the real reference (tools/make_golden.py).
This is what pins the oracle: every fixture must be reproduced byte for byte (minus @PG)."""
import subprocess

import pytest

import harness as H
import oracle as orc


@pytest.mark.parametrize("name", H.SE)
def test_oracle_cli_matches_reference_sam(name, tmp_path):
    fa, fq, _, _ = H.fixture_paths(name)
    out = tmp_path / "o.sam"
    flags = H.MANIFEST[name]["flags"]
    r = subprocess.run([orc.CLI, "-a", fq, "-d", fa] + flags + ["-p", "1", "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = "".join(l for l in open(out) if not l.startswith("@PG"))
    assert got == H.golden_sam(name)


@pytest.mark.parametrize("name", H.PE)
def test_oracle_cli_matches_reference_sam_pe(name, tmp_path):
    fa, fq, fq2, _ = H.fixture_paths(name)
    out = tmp_path / "o.sam"
    r = subprocess.run([orc.CLI, "-a", fq, "-b", fq2, "-d", fa] + H.MANIFEST[name]["flags"] + ["-p", "1", "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = "".join(l for l in open(out) if not l.startswith("@PG"))
    assert got == H.golden_sam(name)


def test_oracle_threads_same_record_set(tmp_path):
    """-p 4 gives the same records (batch order may differ with >50k reads; here one batch)."""
    name = "ct_basic"
    fa, fq, _, _ = H.fixture_paths(name)
    out = tmp_path / "o.sam"
    r = subprocess.run([orc.CLI, "-a", fq, "-d", fa] + H.MANIFEST[name]["flags"] + ["-p", "4", "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = sorted(l for l in open(out) if not l.startswith("@"))
    exp = sorted(l + "\n" for l in H.golden_sam(name).splitlines() if not l.startswith("@"))
    assert got == exp


def test_kat_bit_primitives():
    """Known answers dumped from the reference's own functions."""
    L = orc.lib()
    assert L.orc_myrand(0, 1) == 3753797568
    assert L.orc_myrand(1, 1) == 1753423252
    assert L.orc_myrand(12345, 1) == 2620591974
    import random
    rng = random.Random(7)
    for _ in range(2000):
        x = rng.getrandbits(32)
        digits = [(x >> (2 * i)) & 3 for i in range(16)]
        want = sum((1 if d == 3 else d) * 3 ** i for i, d in enumerate(digits))
        assert L.orc_XT(x) == want
        y = rng.getrandbits(64)
        pairs = [(y >> (2 * i)) & 3 for i in range(32)]
        assert L.orc_XM64(y) == sum(1 for p in pairs if p)
        assert L.orc_XC64(y) == sum((1 if p == 1 else 3) << (2 * i) for i, p in enumerate(pairs))
        assert L.orc_M2_judge(y) == sum((3 if p == 3 else (0 if p == 1 else (p & ((p >> 1) | ((p & 1) << 1))))) << (2 * i)
                                        for i, p in enumerate(pairs))
