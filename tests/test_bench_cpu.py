"""bench.py checks that need no GPU."""
import json
import os

import harness as H


def test_bench_roofline_bytes_of_the_headline_do_not_depend_on_n():
    """The algorithmic bytes per read an N > 1 run prices its launches with (no oracle sample there) are the committed figure of the SAME
    stand-in genome the N = 1 run measures (profiles/algorithmic.json, keyed by genome): `roofline.frac` under --gpus 8 then agrees with the
    N = 1 line instead of being 7.5 x too low."""
    a = json.load(open(os.path.join(H.ROOT, "profiles", "algorithmic.json")))
    assert set(a) >= {"realistic", "uniform"} and a["realistic"]["bytes_per_read"] > 5 * a["uniform"]["bytes_per_read"]
    src = open(os.path.join(H.ROOT, "bench.py")).read()
    assert 'json.load(open(aj))[args.genome]' in src
    recs = sorted(f for f in os.listdir(H.ROOT) if f.startswith("BENCH_r") and f.endswith(".json"))
    line = json.load(open(os.path.join(H.ROOT, recs[-1]))).get("parsed") if recs else None
    if line and "hg38-like" in line["config"]["workload"] and line["n_gpus"] == 1:  # the driver's last N = 1 line measured the same bytes per read
        assert abs(line["config"]["algorithmic_bytes_per_read"] / a["realistic"]["bytes_per_read"] - 1) < 0.02
