"""Searches on small indexes for a kernel trace: the fixed costs of the batch-1 kernels (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
for n in (4096, 65536, 1_000_000):
    idx = dawn.VectorIndex(0); idx.fill_synthetic(1, 0, n, 1)
    Q = synth.unit_rows(2, 0, 1)
    for _ in range(30): idx.search_batch(Q, 10)
    idx.close()
