"""Sweep of the packed stream's dynamic work assignment (dev tool): share of the index handed out on demand (sixteenths, option
i6_dyn_share) x chunk length (sub-tiles, option i6_dyn_chunk) x chunks per wave / shared by the workgroup (option i6_dyn_queue), interleaved rounds on one index, results compared bit for bit
against the static assignment.  python tools/stream_dyn_sweep.py [rows=12500000] [rounds=3]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
idx = dawn.VectorIndex(0)
idx.set_option("i6_min_rows", 100_000)
idx.fill_synthetic(1, 0, rows, 1)
Q = np.concatenate([synth.unit_rows(2, 0, 3), synth.planted_queries(1, [rows - 5], 4)])
idx.set_option("stream_dynamic_tail", 0)
want = [idx.search(q, 20) for q in Q]
combos = [(0, 0, 0), (2, 16, 0)] + [(s, c, 1) for s in (1, 2, 4, 8, 12) for c in (32, 64)]
res = {}
for r in range(rounds):
    for share, chunk, queue in combos:
        idx.set_option("stream_dynamic_tail", 1 if share else 0)
        if share:
            idx.set_option("i6_dyn_share", share)
            idx.set_option("i6_dyn_chunk", chunk)
            idx.set_option("i6_dyn_queue", 1 + queue)
        for q, w in zip(Q, want):
            got = idx.search(q, 20)
            assert np.array_equal(got[0], w[0]) and np.array_equal(got[1].view(np.uint32), w[1].view(np.uint32)), (share, chunk, queue)
        idx.profile_enable(True)
        for i in range(8):
            idx.search(Q[i % 4], 10)
        n, ms = idx.profile_read()
        idx.profile_enable(False)
        res.setdefault((share, chunk, queue), []).append(ms / n * 1e3)
for (share, chunk, queue), v in res.items():
    name = f"share {share:2d}/16 chunk {chunk:3d} {'queue' if queue else 'waves'}" if share else "static                    "
    print(f"rows={rows} {name}: kernel us {[round(x, 1) for x in v]}  best {min(v):8.1f}  {rows * 240.25 / (min(v) * 1e-6) / 8e12:.3f} of 8 TB/s")
print(idx.stats())
