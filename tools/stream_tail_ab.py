"""A/B of the single-query streams' work assignment (dev tool): static for every unit (option stream_dynamic_tail = 0) against the
last eighth handed out on demand (default) — the packed 5-bit stream and the f32-row stream, interleaved rounds on one index,
results compared bit for bit.  python tools/stream_tail_ab.py [rows=100000000] [rounds=4]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
Q = np.concatenate([synth.unit_rows(2, 0, 3), synth.planted_queries(1, [rows - 5], 4)])
want = [idx.search(q, 20) for q in Q]
res = {}
for r in range(rounds):
    for path, b1 in (("packed 5-bit", 1), ("f32 rows", 0)):
        idx.set_option("f16_shadow_b1", b1)
        for mode in (1, 0):
            idx.set_option("stream_dynamic_tail", mode)
            for q, w in zip(Q, want):
                got = idx.search(q, 20)
                assert np.array_equal(got[0], w[0]) and np.array_equal(got[1].view(np.uint32), w[1].view(np.uint32)), (path, mode)
            idx.profile_enable(True)
            for i in range(6):
                idx.search(Q[i % 4], 10)
            n, ms = idx.profile_read()
            idx.profile_enable(False)
            res.setdefault((path, mode), []).append(ms / n * 1e3)
idx.set_option("f16_shadow_b1", 1)
for (path, mode), v in res.items():
    bpr = 240.25 if path.startswith("packed") else 1536.0
    print(f"rows={rows} {path:13s} {'dynamic tail' if mode else 'static      '}: kernel us {[round(x, 1) for x in v]}  best {rows * bpr / (min(v) * 1e-6) / 8e12:.3f} of 8 TB/s")
print(idx.stats())
