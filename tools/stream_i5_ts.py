"""Per-wave timestamps of the packed stream (experiments library: DAWN_LIB=.../libdawn_hip_exp.so; dev tool): when does every one
of the 2048 waves start streaming, finish streaming, finish refining, leave — i.e. how much of a launch is tail imbalance of the
static interleaved assignment.  python tools/stream_i5_ts.py [rows=100000000]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import _lib, synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(2, 0, 4)
names = ["entry", "stream starts", "stream done", "refined", "workgroup done"]
for r in range(6):
    idx.search(Q[r % 4], 10)
    if r < 2:
        continue
    ts = (C.c_ulonglong * (2048 * 5))()
    assert _lib.lib.dawn_debug_read_ts_i6(ts, 2048 * 5) == 0
    t = np.array(list(ts), dtype=np.float64).reshape(2048, 5)
    t = (t - t[:, 0].min()) / 100.0  # us since the first wave entered
    print(f"search {r}: kernel spans {t[:, 4].max():8.1f} us")
    for i, nm in enumerate(names):
        c = t[:, i]
        print(f"   {nm:16s} min {c.min():8.1f}  p10 {np.percentile(c, 10):8.1f}  median {np.median(c):8.1f}  p90 {np.percentile(c, 90):8.1f}  max {c.max():8.1f}")
    sd = t[:, 2]
    byx = [sd[(np.arange(2048) // 8) % 8 == x].mean() for x in range(8)]  # workgroup b = wave // 8 runs on XCD b % 8
    print("   stream done, mean per XCD:", "  ".join(f"{v:8.1f}" for v in byx))
    wg = sd.reshape(256, 8)  # [workgroup][wave of the workgroup]
    print(f"   stream done: spread of the workgroups' means {wg.mean(axis=1).max() - wg.mean(axis=1).min():6.1f} us, mean spread inside a "
          f"workgroup {(wg.max(axis=1) - wg.min(axis=1)).mean():6.1f} us; mean by wave slot: " + " ".join(f"{v:7.1f}" for v in wg.mean(axis=0)))
    if r == 5:
        order = np.argsort(wg.mean(axis=1))
        print("   slowest workgroups:", order[-8:].tolist(), "fastest:", order[:8].tolist())
    dur = t[:, 2] - t[:, 1]
    print(f"   streaming time per wave: min {dur.min():8.1f}  median {np.median(dur):8.1f}  max {dur.max():8.1f}  (max - min = {dur.max() - dur.min():6.1f} us)")
