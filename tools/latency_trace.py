"""Dev tool: searches on a 1M-row index for a kernel trace (rocprofv3 --kernel-trace -- python3 tools/latency_trace.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(2, 0, 256)
for B, it in ((1, 200), (8, 100), (256, 50)):
    idx.search_batch(Q[:B], 10)
    t0 = time.time()
    for _ in range(it):
        idx.search_batch(Q[:B], 10)
    print(f"B={B} host {1e3 * (time.time() - t0) / it:.3f} ms/search", flush=True)
print(idx.stats())
