import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(3, 0, 256)
for B in (128, 129, 160, 192, 256):
    idx.set_option("i8_batched", 0); a = idx.search_batch(Q[:B], 10)
    idx.set_option("i8_batched", 1); b = idx.search_batch(Q[:B], 10)
    bad = [i for i in range(B) if not (np.array_equal(a[0][i], b[0][i]) and np.array_equal(a[1][i], b[1][i]))]
    print("B", B, "differing queries", bad[:40], len(bad))
    for i in bad[:3]:
        print("  f16", a[0][i], a[1][i]); print("  i8 ", b[0][i], b[1][i])
