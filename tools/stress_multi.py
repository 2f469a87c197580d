#!/usr/bin/env python3
"""Stress of `basal -G a,b,..` through the command line: N fresh processes on one golden fixture, each run's SAM against the golden one.
A differing run is reported with its first differing records (and appended to --log); tools/stress_pipe.py is the in-process form.

  python tools/stress_multi.py tdel_pipeline 0,0,0 20000 2000 [--log profiles/r04_stress_multi.log]
"""
import argparse
import gzip
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H  # noqa: E402

BASAL_BIN = os.path.join(ROOT, "basal_amd", "bin", "basal")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("gpus")
    ap.add_argument("pipe_bytes")
    ap.add_argument("iters", type=int)
    ap.add_argument("--log", default=None)
    a = ap.parse_args()
    fa, fq, _, _ = H.fixture_paths(a.name)
    td = tempfile.mkdtemp()
    plain = os.path.join(td, os.path.basename(fq)[:-3])
    open(plain, "wb").write(gzip.open(fq, "rb").read())
    gold = H.golden_sam(a.name)
    env = dict(os.environ, BASAL_PIPE_BYTES=a.pipe_bytes)
    bad, msgs, t0 = 0, [], time.time()
    for i in range(a.iters):
        out = os.path.join(td, "o.sam")
        r = subprocess.run([BASAL_BIN, "-a", plain, "-d", fa] + H.MANIFEST[a.name]["flags"] + ["-p", "4", "-G", a.gpus, "-o", out], capture_output=True, text=True, env=env)
        got = "".join(l for l in open(out) if not l.startswith("@PG")) if r.returncode == 0 else "rc=%d %s" % (r.returncode, r.stderr[-300:])
        if got != gold:
            bad += 1
            g, e = got.split("\n"), gold.split("\n")
            d = [(k, x, y) for k, (x, y) in enumerate(zip(g, e)) if x != y]
            m = "run %d differs: %d lines differ (got %d lines, golden %d)" % (i, len(d), len(g), len(e))
            for k, x, y in d[:3]:
                m += "\n  line %d\n   got: %s\n   exp: %s" % (k, x[:400], y[:400])
            msgs.append(m)
            print(m, flush=True)
        if (i + 1) % 100 == 0:
            print("[stress] %s -G %s: %d bad of %d, %.0f s" % (a.name, a.gpus, bad, i + 1, time.time() - t0), flush=True)
    summary = "%s -G %s, %s-byte batches, through the command line: %d bad of %d runs in %.0f s" % (a.name, a.gpus, a.pipe_bytes, bad, a.iters, time.time() - t0)
    print(summary, flush=True)
    if a.log:
        with open(a.log, "a") as f:
            for m in msgs:
                f.write(m + "\n")
            f.write(summary + "\n")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
