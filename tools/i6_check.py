"""Quick correctness check of the 6-bit stream against the int8 path and the oracle (dev tool)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402
from oracle import oracle_lib as oracle  # noqa: E402

for n in (1, 63, 65, 1000, 100_003, 1_000_000):
    idx = dawn.VectorIndex(0)
    idx.set_option("i6_min_rows", 0)
    idx.fill_synthetic(1, 0, n, 1)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = np.concatenate([synth.unit_rows(2, 0, 3), synth.planted_queries(1, [n // 2], 4)])
    for k in (1, 10, 20, 64):
        for q in Q:
            lab, dist = idx.search(q, k)
            olab, odist = oracle.scan_topk(x, ids, q, k, threads=8)
            assert np.array_equal(lab, olab), (n, k, lab, olab)
            assert np.array_equal(dist.view(np.uint32), odist.view(np.uint32)), (n, k)
    sc, rows = idx.debug_stream_lists(Q[0])
    valid = rows != 0xFFFFFFFF
    got = rows[valid].astype(np.int64)
    exact = x.astype(np.float64) @ Q[0].astype(np.float64)
    diff = sc[valid].astype(np.float64) - exact[got]
    print(n, "ok", idx.stats(), "ub - exact: min %.3g max %.3g median %.3g" % (diff.min(), diff.max(), np.median(diff)), flush=True)
