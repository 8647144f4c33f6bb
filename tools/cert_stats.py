#!/usr/bin/env python3
"""Certificate statistics of the int8 filters on realistic score distributions (option "synth_dist": 0 uniform spec rows,
1 Gaussian, 2 heavy-tailed fixed dims, 3 heavy-tailed per-row dims): per distribution, k and batch size the rate, the
queries whose 64-row certificate failed (second_chances) and the exact passes (fallbacks).
usage: cert_stats.py [rows] [dists, e.g. 0,1,2,3]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dists = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3]
NAMES = {0: "uniform", 1: "gaussian", 2: "heavy_fixed_dims", 3: "heavy_row_dims"}
for dist in dists:
    qi = dawn.VectorIndex(0)
    qi.set_option("synth_dist", dist)
    qi.fill_synthetic(2, 0, 256, 1)
    Q, _ = qi.get_rows(0, 256)
    qi.close()
    idx = dawn.VectorIndex(0)
    idx.set_option("synth_dist", dist)
    t0 = time.time()
    idx.fill_synthetic(1, 0, rows, 1)
    planted, _ = idx.get_rows(4242 % rows, 1)
    Q[0] = planted[0]
    out = {"dist": NAMES[dist], "rows": rows, "fill_s": round(time.time() - t0, 2), "max_abs_q": float(np.abs(Q).max())}
    for k in (10, 20):
        for B in (1, 256):
            s0 = idx.stats()
            n = 64 if B == 1 else 3
            idx.search_batch(Q[:B], k)
            ok = None
            t0 = time.perf_counter()
            for i in range(n):
                if B == 1:
                    lab, dist_, _ = idx.search_batch(Q[i:i + 1], k)
                else:
                    lab, dist_, _ = idx.search_batch(Q, k)
                if i == 0:
                    ok = bool(lab[0][0] == 1 + 4242 % rows)
            el = time.perf_counter() - t0
            s1 = idx.stats()
            nq = n * B + B
            out[f"k{k}_B{B}"] = {"q_per_s": round(n * B / el, 1), "ms_per_call": round(el / n * 1e3, 3),
                                 "second_chance_rate": round((s1["second_chances"] - s0["second_chances"]) / nq, 4), "of_which_deepened": round((s1["deepened"] - s0["deepened"]) / nq, 4),
                                 "fallback_rate": round((s1["fallbacks"] - s0["fallbacks"]) / nq, 4), "planted_ok": ok}
    print(json.dumps(out), flush=True)
    idx.close()
