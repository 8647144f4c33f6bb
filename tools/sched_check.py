import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
for dtype in ("f32", "bf16"):
    for n, B in ((100_003, 33), (20_001, 256), (1_000_000, 256), (3_000_000, 40), (8193, 9), (4097, 256), (64, 16)):
        idx = dawn.VectorIndex(0, dtype=dtype)
        idx.fill_synthetic(1, 0, n, 1)
        Q = synth.unit_rows(2, 0, B)
        idx.set_option("mfma_sched", 0)
        l0, d0, f0 = idx.search_batch(Q, 10)
        for sched in (1, 5):
            idx.set_option("mfma_sched", sched)
            l1, d1, f1 = idx.search_batch(Q, 10)
            ok = np.array_equal(l0, l1) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32)) and np.array_equal(f0, f1)
            print(dtype, n, B, "sched", sched, "OK" if ok else "MISMATCH", idx.stats(), flush=True)
        idx.set_option("mfma_sched", 0)
