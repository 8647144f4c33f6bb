"""Host-side cost of a sharded search (logical shards on one device): issuing threads on / off (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
for G in (1, 4, 8):
    idx = dawn.VectorIndex(devices=[0] * G)
    idx.fill_synthetic(1, 0, 200_000, 1)
    q = synth.unit_rows(2, 0, 1)
    for threads in (1, 0):
        idx.set_option("shard_threads", threads)
        for _ in range(20):
            idx.search_batch(q, 10)
        t0 = time.perf_counter()
        for _ in range(300):
            idx.search_batch(q, 10)
        print(f"G={G} shard_threads={threads}: {(time.perf_counter() - t0) / 300 * 1e3:.3f} ms per search (200 k rows, B = 1)", flush=True)
    idx.close()
