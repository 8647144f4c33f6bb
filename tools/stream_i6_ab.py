"""A/B of the single-query streams (dev tool): the int8 shadow's default kernel against the 6-bit shadow's (scan_i6.hip) over
waves per CU x ring depth, interleaved rounds, results compared bit for bit (labels and distance bits; k = 10 and 20).
python tools/stream_i6_ab.py [rows=100000000] [rounds=3]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 5
os.environ["DAWN_I6_BITS"] = str(bits)
idx = dawn.VectorIndex(0)
idx.set_option("i6_min_rows", 0)
idx.fill_synthetic(1, 0, rows, 1)
print("memory", idx.memory(), flush=True)
Q = np.concatenate([synth.unit_rows(2, 0, 5), synth.planted_queries(1, [rows // 3], 4)])
idx.set_option("i6_shadow", 0)
want = {k: [idx.search(q, k) for q in Q] for k in (10, 20)}
idx.set_option("i6_shadow", 1)
if bits == 6:
    cfgs = [("i8", 0, 0), ("i6", 256, 12), ("i6", 384, 12), ("i6", 512, 12), ("i6", 384, 6), ("i6", 512, 6), ("i6", 512, 4), ("i6", 512, 3)]
else:
    cfgs = [("i8", 0, 0), ("i6", 512, 8), ("i6", 512, 4), ("i6", 256, 8), ("i6", 384, 8)]
acc = {c: [] for c in cfgs}
wall = {c: [] for c in cfgs}
iters = 8 if rows > 10_000_000 else 100
for r in range(rounds):
    for cfg in cfgs:
        kind, t, ring = cfg
        idx.set_option("i6_shadow", int(kind == "i6"))
        if kind == "i6":
            idx.set_option("i6_scan_threads", t)
            idx.set_option("i6_scan_ring", ring)
        for k in (10, 20):
            for q, w in zip(Q, want[k]):
                got = idx.search(q, k)
                assert np.array_equal(got[0], w[0]) and np.array_equal(got[1].view(np.uint32), w[1].view(np.uint32)), (cfg, k)
        idx.profile_enable(True)
        t0 = time.perf_counter()
        for i in range(iters):
            idx.search(Q[i % len(Q)], 10)
        t1 = time.perf_counter()
        n, ms = idx.profile_read()
        idx.profile_enable(False)
        acc[cfg].append(ms / n)
        wall[cfg].append((t1 - t0) / iters * 1e3)
for cfg, v in acc.items():
    kind, t, ring = cfg
    k = min(v)
    bpr = 384.25 if kind == "i8" else (288.25 if bits == 6 else 240.25)
    print(f"{kind} threads={t:4d} ring={ring:2d}  kernel best {k * 1e3:8.1f} us  all {[round(x * 1e3, 1) for x in v]}  "
          f"{rows * bpr / k / 1e6:8.1f} GB/s = {rows * bpr / k / 1e6 / 8000:.3f} of 8 TB/s   per search (host) {min(wall[cfg]):.3f} ms",
          flush=True)
print("bits", bits, idx.stats())
