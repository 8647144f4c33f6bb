"""Dev tool: kernel time of the packed 5-bit stream against index size (f32 index, 25 .. 100 M rows) — slope = the stream's steady-state rate, intercept = what a launch pays besides (prologue, refinement and exact
rescore in the epilogue, tail imbalance).  python tools/stream_i5_slope.py [refine=0]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

refine = int(sys.argv[1]) if len(sys.argv) > 1 else 0
Q = synth.unit_rows(2, 0, 4)
pts = []
idx = dawn.VectorIndex(0)
idx.reserve(100_000_000)  # (growing 75 M -> 100 M rows would hold both generations of the rows at once)
done = 0
for rows in (12_500_000, 25_000_000, 50_000_000, 75_000_000, 100_000_000):
    idx.fill_synthetic(1, done, rows - done, done + 1)
    done = rows
    idx.set_option("i6_scan_threads", 512)
    idx.set_option("i6_scan_ring", 4)
    idx.set_option("i6_refine", refine)
    best = 1e9
    for r in range(4):
        idx.profile_enable(True)
        for i in range(12):
            idx.search(Q[i % 4], 10)
        n, ms = idx.profile_read()
        idx.profile_enable(False)
        best = min(best, ms / n * 1e3)
    pts.append((rows, best))
    print(f"rows={rows} kernel {best:9.1f} us   {rows * 240.25 / best / 1e6:8.1f} GB/s", flush=True)
for (r0, t0), (r1, t1) in zip(pts, pts[1:]):
    slope = (t1 - t0) / (r1 - r0)  # us per row
    print(f"{r0 // 10**6} M -> {r1 // 10**6} M: {240.25 / slope / 1e6:7.1f} GB/s steady state = {240.25 / slope / 1e6 / 8000:.3f} of spec, "
          f"intercept {t0 - slope * r0:7.1f} us")
print(idx.stats())
