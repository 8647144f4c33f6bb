"""Repeated batch-1 searches must return identical results (dev tool; run from the repo root on a GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
for n in (300_000, 1_000_000, 5_000_000):
    idx = dawn.VectorIndex(0); idx.fill_synthetic(1, 0, n, 1)
    Q = synth.unit_rows(2, 0, 64)
    ref = [idx.search_batch(Q[i:i + 1], 10) for i in range(64)]
    bad = 0
    for rep in range(40):
        for i in range(64):
            l, d, f = idx.search_batch(Q[i:i + 1], 10)
            if not (np.array_equal(l, ref[i][0]) and np.array_equal(d.view(np.uint32), ref[i][1].view(np.uint32))):
                bad += 1
    print("rows", n, "mismatches in 2560 repeated searches:", bad, idx.stats())
