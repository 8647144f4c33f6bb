"""Certificate statistics of the packed 5-bit stream (dev tool): Q random queries per distribution (uniform / Gaussian /
heavy-tailed rows, option synth_dist) and k on a large index, every answer compared bit for bit with the int8 stream's (which the
test suite holds against the oracle), exact passes counted.  python tools/packed_cert_stats.py [rows=100000000] [queries=300]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for dist, name in ((0, "uniform"), (1, "gaussian"), (2, "heavy-tailed, 4 fixed dims x5"), (3, "heavy-tailed, 4 dims per row x5")):
    idx = dawn.VectorIndex(0)
    idx.set_option("synth_dist", dist)
    idx.fill_synthetic(1, 0, rows, 1)
    qi = dawn.VectorIndex(0)
    qi.set_option("synth_dist", dist)
    qi.fill_synthetic(7, 0, nq, 1)
    Q = qi.get_rows(0, nq)[0]
    qi.close()
    for k in (10, 20, 32):
        idx.set_option("i6_shadow", 0)
        want = [idx.search(q, k) for q in Q]
        idx.set_option("i6_shadow", 1)
        s0 = idx.stats()
        t0 = time.perf_counter()
        got = [idx.search(q, k) for q in Q]
        el = time.perf_counter() - t0
        s1 = idx.stats()
        same = all(np.array_equal(g[0], w[0]) and np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32)) for g, w in zip(got, want))
        print(f"rows={rows} {name:32s} k={k:2d}: {nq} queries, identical to the int8 stream: {same}, exact passes "
              f"{s1['fallbacks'] - s0['fallbacks']}, {el / nq * 1e3:.3f} ms per search", flush=True)
    idx.close()
