"""Dev tool: batched pass time vs the candidate target of the sampled thresholds: python tools/target_sweep.py [rows] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(2, 0, B)
ref = None
for target in [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "4096,3072,2048,1536,1024,512".split(","))]:
    idx.set_option("mfma_target", target)
    out = idx.search_batch(Q, 20)
    ref = ref or out
    same = bool(np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1]))
    ms = idx.debug_time_full_pass(B, 8)
    t0 = time.time()
    for _ in range(5):
        idx.search_batch(Q, 20)
    print(f"rows={rows} B={B} target={target:5d} full pass {ms*1e3:8.1f} us  search {(time.time()-t0)/5*1e3:7.3f} ms same={same} {idx.stats()}", flush=True)
