"""Dev tool: int8 vs f16 shadow on the matrix-core (batched) path: python tools/i8_batched_check.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(3, 0, 256)
Q[7] = synth.planted_queries(1, [rows // 2], 5)[0]
it = 20 if rows <= 20_000_000 else 5
for k in (10, 20):
    for B in (4, 33, 64, 100, 256):
        res = {}
        for i8 in (0, 1):
            idx.set_option("i8_batched", i8)
            out = idx.search_batch(Q[:B], k)
            idx.profile_enable(True)
            t0 = time.time()
            for _ in range(it):
                idx.search_batch(Q[:B], k)
            ms = (time.time() - t0) / it * 1e3
            n, kms = idx.profile_read()
            idx.profile_enable(False)
            res[i8] = (out, ms, kms / max(n, 1))
        same = bool(np.array_equal(res[0][0][0], res[1][0][0]) and np.array_equal(res[0][0][1], res[1][0][1]))
        print(f"rows={rows} k={k} B={B:3d}  f16 {res[0][1]:8.3f} ms (pass {res[0][2]:7.3f})  i8 {res[1][1]:8.3f} ms (pass {res[1][2]:7.3f})  identical={same}  {idx.stats()}", flush=True)
