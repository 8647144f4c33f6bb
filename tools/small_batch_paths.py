"""Dev tool: batches of 2..8 queries through the streaming filter vs the matrix-core path: python tools/small_batch_paths.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(2, 0, 16)
it = 50 if rows <= 20_000_000 else 10
for B in (1, 2, 3, 4, 6, 8, 12, 16):
    res = []
    for mmb in (100000, 2):
        if B == 1 and mmb == 2:
            mmb = 1
        idx.set_option("mfma_min_batch", mmb)
        idx.search_batch(Q[:B], 10)
        t0 = time.time()
        for _ in range(it):
            idx.search_batch(Q[:B], 10)
        res.append((time.time() - t0) / it * 1e3)
    print(f"rows={rows} B={B:2d}  stream {res[0]:8.3f} ms   matrix-core {res[1]:8.3f} ms", flush=True)
print(idx.stats())
