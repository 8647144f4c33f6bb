"""A/B of the batch-1 int8 stream kernels (dev tool): the round-2 kernel (shadow_scan_unroll 3) against the software-pipelined
one (6: chip-wide window, 7: per-XCD ranges) at 2 / 4 / 8 waves per CU, interleaved rounds, results compared bit for bit.
python tools/stream_pipe_ab.py [rows=100000000] [rounds=3]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
Q = np.concatenate([synth.unit_rows(2, 0, 3), synth.planted_queries(1, [rows // 3], 4)])
want = [idx.search(q, 20) for q in Q]
cfgs = [(3, 256), (3, 512), (6, 128), (8, 256), (8, 512)]
acc = {c: [] for c in cfgs}
iters = 8 if rows > 10_000_000 else 100
for r in range(rounds):
    for (u, t) in cfgs:
        idx.set_option("shadow_scan_unroll", u)
        idx.set_option("shadow_scan_threads", t)
        idx.set_option("shadow_scan_blocks", 256)
        for q, w in zip(Q, want):
            got = idx.search(q, 20)
            assert np.array_equal(got[0], w[0]) and np.array_equal(got[1].view(np.uint32), w[1].view(np.uint32)), (u, t)
        idx.profile_enable(True)
        for i in range(iters):
            idx.search(Q[i % len(Q)], 10)
        n, ms = idx.profile_read()
        idx.profile_enable(False)
        acc[(u, t)].append(ms / n)
for (u, t), v in acc.items():
    k = min(v)
    print(f"unroll={u} threads={t:4d}  kernel best {k * 1e3:8.1f} us  all {[round(x * 1e3, 1) for x in v]}  "
          f"{rows * 384.25 / k / 1e6:8.1f} GB/s = {rows * 384.25 / k / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
print(idx.stats())
