"""Cost of the reference's one-row-per-call insert path (search_provider.rs:127-153,280-284): dawn_index_add per call, and a
fill_index_from_db-shaped loop of 100 k rows against one dawn_index_add_batch.  python tools/add_bench.py [rows=100000]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402


def run(n=100_000, device=0):
    rows = synth.unit_rows(5, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    idx = dawn.VectorIndex(device)
    idx.reserve(n)
    t0 = time.perf_counter()
    for i in range(n):
        idx.add(int(ids[i]), rows[i])
    idx.search(rows[0], 1)  # (the last staged rows join the index here)
    t_loop = time.perf_counter() - t0
    idx2 = dawn.VectorIndex(device)
    t0 = time.perf_counter()
    idx2.add_batch(ids, rows)
    t_batch = time.perf_counter() - t0
    same = bool(np.array_equal(idx.search(rows[n // 2], 10)[0], idx2.search(rows[n // 2], 10)[0]))
    # the insert path of a running node: one add, then a search (every add is flushed on its own)
    t0 = time.perf_counter()
    for i in range(200):
        idx.add(10_000_000 + i, rows[i])
        idx.search(rows[i], 1)
    t_pair = (time.perf_counter() - t0) / 200
    return {"rows": n, "add_loop_us_per_row": t_loop / n * 1e6, "add_loop_rows_per_s": n / t_loop,
            "add_batch_rows_per_s": n / t_batch, "add_then_search_us": t_pair * 1e6, "same_results": same,
            "note": "python ctypes call overhead (~1.5 us) is inside the per-row figure"}


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 100_000)))
