"""Dev tool: int8 shadow vs f16 shadow on the streaming path: python tools/i8_check.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(2, 0, 64)
it = 50 if rows <= 20_000_000 else 10
ref = None
for i8 in (0, 1):
    idx.set_option("i8_shadow", i8)
    for k in (10, 20, 64):
        for B in (1, 2, 3):
            out = [idx.search_batch(Q[j:j + B], k) for j in range(0, 24, B)]
            idx.search_batch(Q[:B], k)
            t0 = time.time()
            for _ in range(it):
                idx.search_batch(Q[:B], k)
            ms = (time.time() - t0) / it * 1e3
            lab = np.concatenate([o[0] for o in out]); dist = np.concatenate([o[1] for o in out])
            key = (k,)
            if i8 == 0 and B == 1:
                ref = ref or {}
                ref[k] = (lab, dist)
            same = bool(np.array_equal(lab, ref[k][0]) and np.array_equal(dist, ref[k][1]))
            print(f"rows={rows} i8={i8} k={k} B={B} {ms:8.3f} ms  identical_to_f16_B1={same} stats={idx.stats()}", flush=True)
