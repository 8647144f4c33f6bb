"""Dev tool: which share of a topical batch ends in the ladder as a function of the candidate target of the sampled thresholds
(option "mfma_target"): overflowing candidate buffers (too low a threshold inside a dense shell of near-ties) or certificates that
fail (too high a threshold)?  python tools/topical_target_sweep.py [rows=25000000] [dist=4]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
dist = int(sys.argv[2]) if len(sys.argv) > 2 else 4
QROW0 = 1 << 40
qi = dawn.VectorIndex(0)
qi.set_option("synth_dist", dist)
qi.fill_synthetic(1, QROW0, 256 * 256, 1)
Q = qi.get_rows(0, 256 * 256)[0][::256].copy()
qi.close()
idx = dawn.VectorIndex(0)
idx.set_option("synth_dist", dist)
idx.fill_synthetic(1, 0, rows, 1)
ref = None
for target in (4096, 2048, 1024, 512, 256, 128, 64):
    idx.set_option("mfma_target", target)
    out = idx.search_batch(Q, 10)
    ref = ref or out
    same = bool(np.array_equal(out[0], ref[0]) and np.array_equal(out[1].view(np.uint32), ref[1].view(np.uint32)))
    s0 = idx.stats()
    t0 = time.time()
    for _ in range(3):
        idx.search_batch(Q, 10)
    el = (time.time() - t0) / 3 * 1e3
    s1 = idx.stats()
    r = {kk: round((s1[kk] - s0[kk]) / (3 * 256), 3) for kk in ("second_chances", "deepened", "bounded", "fallbacks")}
    print(f"rows={rows} dist={dist} target={target:5d}: {el:8.2f} ms per batch of 256 (host API); {r}; same answers: {same}", flush=True)
