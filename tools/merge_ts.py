"""Phase timestamps of merge_rescore_kernel (experiments library: DAWN_LIB=.../libdawn_hip_exp.so; dev tool)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth, _lib

names = ["start", "select: lists merged in the waves", "select: block merge done", "rescoring: rows known", "rows gathered", "exact dots done",
         "sorted + certificate", "results written"]
for n in (4096, 1_000_000):
    idx = dawn.VectorIndex(0); idx.fill_synthetic(1, 0, n, 1)
    Q = synth.unit_rows(2, 0, 1)
    acc = np.zeros(8)
    reps = 20
    for _ in range(reps):
        idx.search_batch(Q, 10)
        ts = (C.c_ulonglong * 16)()
        assert _lib.lib.dawn_debug_read_ts(ts, 16) == 0
        t = np.array(list(ts)[:8], dtype=np.float64)
        acc += (t - t[0]) * 10.0 / 1000.0  # 100-MHz ticks -> us
    print("rows", n)
    prev = 0.0
    for name, v in zip(names, acc / reps):
        print(f"   {name:40s} at {v:7.2f} us  (+{v - prev:6.2f})")
        prev = v
