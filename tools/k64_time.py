import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, dawnsearch_amd as dawn
from dawnsearch_amd import synth
rows = int(sys.argv[1])
idx = dawn.VectorIndex(0); idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(2, 0, 256)
for B in (1, 256):
    for k in (10, 64):
        idx.search_batch(Q[:B], k)
        t0 = time.time()
        for _ in range(10): idx.search_batch(Q[:B], k)
        print(f"rows={rows} B={B} k={k}: {(time.time()-t0)/10*1e3:.3f} ms", idx.stats(), flush=True)
