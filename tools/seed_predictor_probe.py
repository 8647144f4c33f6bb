"""Would the seed search predict a packed-certificate failure?  (round-4 verdict item 7: per-query demotion on the device.)  For queries on
100 M topical rows: does the packed stream's certificate fail on the full index, and does it fail on the first 1/32 of the rows (what the
bounded pass's seed search sees)?  Recall / precision of "seed fails" as a predictor of "full fails".  python tools/seed_predictor_probe.py [rows] [dist]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dist = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nq = 192
qi = dawn.VectorIndex(0)
qi.set_option("synth_dist", dist)
qi.fill_synthetic(1, 1 << 40, nq * 256, 1)
Q = qi.get_rows(0, nq * 256)[0][::256].copy()
qi.close()


def failures(n):
    idx = dawn.VectorIndex(0)
    idx.set_option("synth_dist", dist)
    idx.set_option("f6_shadow", 0)
    idx.set_option("i6_min_rows", 0)
    idx.fill_synthetic(1, 0, n, 1)
    idx.set_option("ladder_feedback", 0)  # always the packed stream first
    out = []
    for q in Q:
        s0 = idx.stats()["packed_failures"]
        idx.search(q, 10)
        out.append(idx.stats()["packed_failures"] - s0)
    idx.close()
    return np.array(out, dtype=bool)


full = failures(rows)
for shift in (5, 4, 3):
    seed = failures(rows >> shift)
    tp = int(np.sum(full & seed)); fn = int(np.sum(full & ~seed)); fp = int(np.sum(~full & seed))
    print(f"rows={rows} dist={dist}: full index fails {int(full.sum())} of {nq}; seed over 1/{1 << shift} of the rows fails {int(seed.sum())}; "
          f"recall {tp / max(tp + fn, 1):.2f} precision {tp / max(tp + fp, 1):.2f}", flush=True)
