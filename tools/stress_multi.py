import os, sys, subprocess, gzip, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H
BASAL_BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "basal_amd", "bin", "basal")
name = sys.argv[1]; gpus = sys.argv[2]; pb = sys.argv[3]; N = int(sys.argv[4])
fa, fq, _, _ = H.fixture_paths(name)
td = tempfile.mkdtemp()
plain = os.path.join(td, os.path.basename(fq)[:-3])
open(plain, "wb").write(gzip.open(fq, "rb").read())
gold = H.golden_sam(name)
env = dict(os.environ, BASAL_PIPE_BYTES=pb)
bad = 0
for i in range(N):
    out = os.path.join(td, "o.sam")
    r = subprocess.run([BASAL_BIN, "-a", plain, "-d", fa] + H.MANIFEST[name]["flags"] + ["-p", "4", "-G", gpus, "-o", out], capture_output=True, text=True, env=env)
    got = "".join(l for l in open(out) if not l.startswith("@PG")) if r.returncode == 0 else "rc=%d %s" % (r.returncode, r.stderr[-300:])
    if got != gold:
        bad += 1
        g, e = got.split("\n"), gold.split("\n")
        d = [(k, a, b) for k, (a, b) in enumerate(zip(g, e)) if a != b]
        print("run %d differs: %d lines differ (got %d lines, golden %d); first:" % (i, len(d), len(g), len(e)))
        for k, a, b in d[:2]:
            print("  line", k, "\n   got:", a[:200], "\n   exp:", b[:200])
print("%s -G %s bytes %s: %d bad of %d" % (name, gpus, pb, bad, N))
