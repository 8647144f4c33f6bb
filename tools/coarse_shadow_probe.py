"""What would a coarser filter shadow cost the certificate?  Quantise the int8 shadow to fewer levels ("debug_i8_levels": 31 = a
6-bit shadow's error, 63 = 7-bit) with the bytes unchanged, and look at certificate statistics and time per query at batch 1
(dev tool).  python tools/coarse_shadow_probe.py [rows=100000000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(2, 0, 24)
want = {k: [idx.search(q, k) for q in Q[:6]] for k in (10, 20)}
for levels in (127, 63, 31, 15):
    idx.set_option("debug_i8_levels", levels)
    for k in (10, 20):
        s0 = idx.stats()
        for i, q in enumerate(Q[:6]):
            got = idx.search(q, k)
            assert np.array_equal(got[0], want[k][i][0]) and np.array_equal(got[1].view(np.uint32), want[k][i][1].view(np.uint32))
        t0 = time.perf_counter()
        for q in Q:
            idx.search(q, k)
        ms = (time.perf_counter() - t0) / len(Q) * 1e3
        s1 = idx.stats()
        n = s1["searches"] - s0["searches"]
        print(f"levels {levels:3d} k={k:2d}: {ms:7.3f} ms per query; of {n} searches: second chances {s1['second_chances'] - s0['second_chances']}, "
              f"settled by deepening {s1['deepened'] - s0['deepened']}, exact passes {s1['fallbacks'] - s0['fallbacks']}", flush=True)
